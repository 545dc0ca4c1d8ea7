cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_alpha_gpu.py tests/test_two_ranks_gpu.py tests/test_config2_full_gpu.py tests/test_api_gpu.py tests/test_end_to_end_gpu.py -q -x 2>&1 | tail -15
python - <<'PY'
import sys, time, torch
sys.path[:0] = [".", "anomaly-clustering_amd/models", "tests"]
import acnative
from patchcore.pipeline import HipOps, run_category
dev = torch.device("cuda:0")
ops = HipOps(core=None); ops.embed = lambda x: x.contiguous()
g = torch.Generator().manual_seed(0)
Z = torch.randn(83, 784, 4096, generator=g).to(dev); B = torch.randn(83, 784, 4096, generator=g).to(dev)
for prune in (True, False, True, False):
    torch.cuda.synchronize(); t = time.perf_counter()
    out = run_category(ops, Z, "supervised", tau=1.0, train_images=B, train_ratio=1.0, prune=prune, want_pdist=False, want_host=False)
    torch.cuda.synchronize(); print("prune", prune, "%.1f ms" % (1e3 * (time.perf_counter() - t)))
PY
