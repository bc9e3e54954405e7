cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out
( time timeout 1200 python -m pytest tests -m gpu -q -x ) > gpurun_out/gpu_tests.log 2>&1
( time timeout 300 python -c "import __graft_entry__ as g; g.smoke()" ) > gpurun_out/smoke.log 2>&1
( time timeout 600 python bench.py --no-cpu-baseline ) > gpurun_out/bench_default.log 2>&1
tail -3 gpurun_out/gpu_tests.log; tail -4 gpurun_out/smoke.log; tail -6 gpurun_out/bench_default.log | cut -c1-600
