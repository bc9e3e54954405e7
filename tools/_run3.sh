cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r1p -o bench -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/prof_r1p.log 2>&1
tail -1 gpurun_out/prof_r1p.log | cut -c1-400
