"""MI355X-native drop-in for the reference's `patchcore` package (hot path only).

Use exactly like the reference: ``PYTHONPATH=<...>/anomaly-clustering_amd/models`` and
``import patchcore.backbones, patchcore.common, patchcore.sampler, patchcore.utils``.
"""
