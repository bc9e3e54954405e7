#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_training_trajectory_gpu.py tests/test_train_step_gpu.py -q -m gpu -x > gpurun_out/c1_tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/c1_tests.log
tail -3 gpurun_out/c1_tests.log
timeout 600 python tools/ab_tail.py > gpurun_out/c1_ab_tail.log 2>&1; echo "ab rc=$?"
grep variant gpurun_out/c1_ab_tail.log
for cfg in "cfg5 --size 1024 -K 200 -N 64 --batch 8" "cfg2 --gan proggan --size 1024 -K 64 -N 16 --batch 32"; do
  set -- $cfg; name=$1; shift
  timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_c1_$name -o bench -- python bench.py "$@" --steps 4 --warmup 2 --no-cpu-baseline --no-extra --no-product-run --no-direct-run --single-stream --precision auto --extra-out gpurun_out/prof_c1_${name}_extra.json > gpurun_out/prof_c1_$name.log 2>&1
  python tools/prof_summary.py gpurun_out/prof_c1_$name 11 > gpurun_out/c1_step_${name}_kernel_stats.md
  python tools/phase_breakdown.py gpurun_out/prof_c1_$name 7 > gpurun_out/c1_phases_${name}.md 2>&1
  find gpurun_out/prof_c1_$name -name "*kernel_trace.csv" -delete
  head -30 gpurun_out/c1_step_${name}_kernel_stats.md | cut -c1-150
done
