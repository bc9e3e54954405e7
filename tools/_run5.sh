cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out
( time timeout 600 python bench.py ) > gpurun_out/bench_default.log 2>&1
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r1m -o bench -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/prof_r1m.log 2>&1
tail -5 gpurun_out/bench_default.log | cut -c1-200
