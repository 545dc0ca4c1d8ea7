cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2a
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r2a/tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2a/tests.log
tail -25 gpurun_out/r2a/tests.log
timeout 600 python bench.py > gpurun_out/r2a/bench2.json 2> gpurun_out/r2a/bench2.err; tail -c 1500 gpurun_out/r2a/bench2.json; tail -3 gpurun_out/r2a/bench2.err
timeout 600 python bench.py --config 5b --n 1200 --no-cpu-baseline > gpurun_out/r2a/bench5b_1200.json 2> gpurun_out/r2a/bench5b.err; tail -c 1200 gpurun_out/r2a/bench5b_1200.json; tail -3 gpurun_out/r2a/bench5b.err
