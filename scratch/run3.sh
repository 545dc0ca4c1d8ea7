cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2c
timeout 2400 python -m pytest tests -m gpu -q > gpurun_out/r2c/tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2c/tests.log
tail -60 gpurun_out/r2c/tests.log
python - <<'PY'
import sys, time, numpy as np, torch
sys.path.insert(0, "anomaly-clustering_amd/models")
import acnative
from scipy.cluster import hierarchy
dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
for n in (167, 2000, 10000):
    X = torch.from_numpy(rng.normal(size=(n, 64)).astype(np.float32)).to(dev)
    d = acnative.pdist(X); torch.cuda.synchronize()
    acnative.ward_linkage(d, n); torch.cuda.synchronize()
    t = time.perf_counter(); Z = acnative.ward_linkage(d, n); torch.cuda.synchronize(); tg = time.perf_counter() - t
    dh = d.cpu().numpy()
    t = time.perf_counter(); ref = hierarchy.linkage(dh, "ward"); ts = time.perf_counter() - t
    print("ward n=%d: gpu %.4f s  scipy %.4f s  identical=%s" % (n, tg, ts, np.array_equal(Z.cpu().numpy(), ref)))
PY
