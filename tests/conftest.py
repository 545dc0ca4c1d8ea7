import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MODELS = os.path.join(ROOT, "anomaly-clustering_amd", "models")
EXAMPLES = os.path.join(ROOT, "anomaly-clustering_amd", "examples")
GOLDEN = os.path.join(ROOT, "tests", "golden")
for p in (ROOT, MODELS, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


# the tests run the seeded reference initialisation (no checkpoint offline): explicit opt-in, see backbones.py
os.environ.setdefault("AC_BACKBONE_SEED", "0")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _have_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _have_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def golden():
    def load(name):
        return np.load(os.path.join(GOLDEN, name), allow_pickle=False)
    return load


@pytest.fixture(scope="session")
def dev():
    import torch
    return torch.device("cuda:0")
