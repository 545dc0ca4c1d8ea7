cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_aggregate_gpu.py tests/test_vit_gpu.py tests/test_api_gpu.py -q 2>&1 | tail -5
for i in 1 2; do timeout 100 python tests/perf_probe.py agg 83 2>/dev/null | tail -1; done
timeout 300 python bench.py --no-cpu-baseline --no-extras --steps 5 2>/dev/null | python -c "import sys,json; d=json.load(sys.stdin); print(d['value'], d['ms_per_step'])"
