cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_vit_gpu.py tests/test_gemm_generations_gpu.py -q 2>&1 | tail -3
for i in 1 2 3; do timeout 100 python tests/perf_probe.py vit 83 2>/dev/null | tail -1; done
