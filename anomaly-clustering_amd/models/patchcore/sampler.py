"""Feature samplers.  Only the identity sampler is on the anomaly-clustering path
(reference: models/patchcore/sampler.py:9-13, constructed at examples/main.py:202);
the coreset samplers belong to PatchCore anomaly *detection* and are out of scope."""


class IdentitySampler:
    def run(self, features):
        return features
