#!/bin/bash
# round-3 first GPU pass: GPU test suite, default bench line, rocprofv3 kernel tables of the fp32 and the default-arithmetic step
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -q -m gpu -x > gpurun_out/r3a_tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3a_tests.log
tail -5 gpurun_out/r3a_tests.log
timeout 900 python bench.py > gpurun_out/r3a_bench.log 2>&1; echo "bench rc=$?"
grep '^{' gpurun_out/r3a_bench.log | tail -1 > gpurun_out/r3a_bench.json
for mode in fp32 auto; do
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r3a_$mode -o bench -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extra --no-product-run --single-stream --precision $mode > gpurun_out/prof_r3a_$mode.log 2>&1
  python tools/prof_summary.py gpurun_out/prof_r3a_$mode 12 > gpurun_out/r3a_step_${mode}_kernel_stats.md
  find gpurun_out/prof_r3a_$mode -name "*kernel_trace.csv" -delete
  head -12 gpurun_out/r3a_step_${mode}_kernel_stats.md | cut -c1-160
done
