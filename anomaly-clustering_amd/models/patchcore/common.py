"""Feature tap, projection modules and constructible shims
(reference: models/patchcore/common.py).

Hot path: NetworkFeatureAggregator / ForwardHook / LastLayerToExtractReachedException
(:211-292) keep their hook + early-exit semantics; Preprocessing / MeanMapper / Aggregator
(:145-183) keep their constructor attributes (``input_dims``, ``output_dim``,
``preprocessing_dim``, ``target_dim``) because AnomalyClusteringCore reads them, but the
arithmetic of all three is fused into one HIP kernel (``acnative.patch_aggregate``) that
AnomalyClusteringCore._embed calls; calling the modules individually also goes through
that kernel.

FaissNN / NearestNeighbourScorer / RescaleSegmentor are only *constructed* on this path
(examples/main.py:237, patchcore.py:327-333): they are attribute-holding shims here, and
their detection methods raise.
"""
import copy

import torch

import acnative


class _DetectionOnly:
    def _refuse(self, *a, **k):
        raise NotImplementedError(
            "%s is PatchCore anomaly-detection machinery; the anomaly-clustering path only constructs it"
            % type(self).__name__)


class FaissNN(_DetectionOnly):
    def __init__(self, on_gpu: bool = False, num_workers: int = 4) -> None:
        self.on_gpu = on_gpu
        self.num_workers = num_workers
        self.search_index = None

    fit = run = save = load = reset_index = _DetectionOnly._refuse


class ApproximateFaissNN(FaissNN):
    pass


class NearestNeighbourScorer(_DetectionOnly):
    def __init__(self, n_nearest_neighbours: int, nn_method=None) -> None:
        self.n_nearest_neighbours = n_nearest_neighbours
        self.nn_method = nn_method if nn_method is not None else FaissNN(False, 4)

    fit = predict = save = load = _DetectionOnly._refuse


class RescaleSegmentor(_DetectionOnly):
    def __init__(self, device, target_size=224):
        self.device = device
        self.target_size = target_size
        self.smoothing = 4

    convert_to_segmentation = _DetectionOnly._refuse


def _as_rows(features):
    """[n, ...] -> fp32 device matrix [n, L] (the flattening MeanMapper / Aggregator apply)."""
    return features.reshape(len(features), -1).to(torch.float32).contiguous()


class MeanMapper(torch.nn.Module):
    """adaptive_avg_pool1d of each flattened row to `preprocessing_dim` (reference :163-170)."""

    def __init__(self, preprocessing_dim):
        super().__init__()
        self.preprocessing_dim = preprocessing_dim

    def forward(self, features):
        return acnative.adaptive_pool_rows(_as_rows(features), self.preprocessing_dim)


class Preprocessing(torch.nn.Module):
    def __init__(self, input_dims, output_dim):
        super().__init__()
        self.input_dims = input_dims
        self.output_dim = output_dim
        self.preprocessing_modules = torch.nn.ModuleList([MeanMapper(output_dim) for _ in input_dims])

    def forward(self, features):
        return torch.stack([m(f) for m, f in zip(self.preprocessing_modules, features)], dim=1)


class Aggregator(torch.nn.Module):
    """Flatten the layer axis and adaptive-average-pool to `target_dim` (reference :173-183)."""

    def __init__(self, target_dim):
        super().__init__()
        self.target_dim = target_dim

    def forward(self, features):
        return acnative.adaptive_pool_rows(_as_rows(features), self.target_dim)


class LastLayerToExtractReachedException(Exception):
    pass


class ForwardHook:
    def __init__(self, hook_dict, layer_name: str, last_layer_to_extract: str):
        self.hook_dict = hook_dict
        self.layer_name = layer_name
        self.raise_exception_to_break = copy.deepcopy(layer_name == last_layer_to_extract)

    def __call__(self, module, input, output):
        self.hook_dict[self.layer_name] = output
        if self.raise_exception_to_break:
            raise LastLayerToExtractReachedException()
        return None


def _resolve(backbone, name):
    """'blocks.10' -> backbone.blocks[10]; 'features.denseblock2' -> submodule; 'layer2' -> child."""
    children = backbone.__dict__["_modules"]
    if "." not in name:
        return children[name]
    block, idx = name.split(".")
    layer = children[block]
    return layer[int(idx)] if idx.isnumeric() else layer.__dict__["_modules"][idx]


class NetworkFeatureAggregator(torch.nn.Module):
    """Runs the backbone only as far as the last requested layer and returns the tapped outputs."""

    def __init__(self, backbone, layers_to_extract_from, device):
        super().__init__()
        self.layers_to_extract_from = layers_to_extract_from
        self.backbone = backbone
        self.device = device
        if not hasattr(backbone, "hook_handles"):
            self.backbone.hook_handles = []
        for handle in self.backbone.hook_handles:
            handle.remove()
        self.outputs = {}
        for name in layers_to_extract_from:
            hook = ForwardHook(self.outputs, name, layers_to_extract_from[-1])
            layer = _resolve(backbone, name)
            target = layer[-1] if isinstance(layer, torch.nn.Sequential) else layer
            self.backbone.hook_handles.append(target.register_forward_hook(hook))
        self.to(self.device)

    def forward(self, images):
        self.outputs.clear()
        with torch.no_grad():
            try:
                _ = self.backbone(images)
            except LastLayerToExtractReachedException:
                pass
        return self.outputs

    def feature_dimensions(self, input_shape):
        """shape[1] of each tapped output for a ones input (token count for ViTs, as upstream)."""
        _input = torch.ones([1] + list(input_shape)).to(self.device)
        _output = self(_input)
        return [_output[layer].shape[1] for layer in self.layers_to_extract_from]
