cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2b
timeout 2400 python -m pytest tests -m gpu -q > gpurun_out/r2b/tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2b/tests.log
tail -60 gpurun_out/r2b/tests.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
