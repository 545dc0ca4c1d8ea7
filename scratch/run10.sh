cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_cluster_gpu.py tests/test_api_gpu.py tests/test_bench_gpu.py tests/test_config2_full_gpu.py -q 2>&1 | tail -8
