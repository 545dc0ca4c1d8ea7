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


# ---- the one-minute CPU-oracle leg of the config-2 test runs BESIDE the GPU tests (tests/_config2_oracle.py)
_CONFIG2_TEST = "test_config2_whole_alpha_matrix_against_oracle"


def pytest_collection_finish(session):
    """GPU session that will run the config-2 test: start its CPU oracle now, as a child process that never touches the GPU."""
    if not _have_gpu() or not any(_CONFIG2_TEST in it.nodeid for it in session.items):
        return
    import subprocess
    import tempfile
    out = os.path.join(tempfile.mkdtemp(prefix="ac_config2_oracle_"), "oracle.npz")
    env = dict(os.environ, HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="", ROCR_VISIBLE_DEVICES="")
    proc = subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_config2_oracle.py"), out], env=env,
                            stdout=subprocess.DEVNULL, stderr=open(out + ".err", "wb"))
    session.config._ac_config2_oracle = (proc, out)


def pytest_unconfigure(config):
    h = getattr(config, "_ac_config2_oracle", None)
    if h is not None and h[0].poll() is None:                    # the test never ran (-x after a failure): end OUR child, by handle
        h[0].kill()


@pytest.fixture(scope="session")
def config2_oracle(request):
    """() -> dict of the oracle's arrays for config 2 (waits for the child started at collection time), or None when no
    child was started or it failed - the test then computes the oracle itself."""
    def get():
        h = getattr(request.config, "_ac_config2_oracle", None)
        if h is None:
            return None
        proc, out = h
        try:
            proc.wait(timeout=300)       # (it takes ~1-2 min beside the GPU tests; the test falls back to computing it itself)
        except Exception:
            proc.kill()
            return None
        if proc.returncode != 0 or not os.path.exists(out):
            sys.stderr.write("config-2 oracle child failed (rc %s): %s\n"
                             % (proc.returncode, open(out + ".err", "rb").read().decode(errors="replace")[-2000:]))
            return None
        return dict(np.load(out, allow_pickle=False))
    return get


@pytest.fixture(scope="session")
def golden():
    def load(name):
        return np.load(os.path.join(GOLDEN, name), allow_pickle=False)
    return load


@pytest.fixture(scope="session")
def dev():
    import torch
    return torch.device("cuda:0")
