cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gemm_generations_gpu.py tests/test_rccl_smoke_gpu.py -q 2>&1 | tail -5
for rep in 1 2; do for p in 2 3 1; do echo -n "policy $p: "; AC_POLICY=$p timeout 120 python tests/perf_probe.py vit 83 2>/dev/null | tail -1; done; done
for p in 2 3; do echo -n "policy $p B=11: "; AC_POLICY=$p timeout 120 python tests/perf_probe.py vit 11 2>/dev/null | tail -1; done
