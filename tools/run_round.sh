#!/bin/bash
# usage: bash tools/run_round.sh <tag> [notests|tests] [lite] — GPU test suite, default bench line (+ side file), rocprofv3 kernel tables and phase
# breakdowns of the fp32w / fp32 / auto step, PMC passes of the dominant conv kernels, host-enqueue measurement with 8 processes
TAG=${1:-r5}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
if [ "$2" != "notests" ]; then
  timeout 2400 python -m pytest tests -q -m gpu -x > gpurun_out/${TAG}_tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/${TAG}_tests.log
  tail -5 gpurun_out/${TAG}_tests.log
  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
fi
timeout 900 python bench.py --extra-out gpurun_out/${TAG}_bench_extra.json > gpurun_out/${TAG}_bench.log 2>&1; echo "bench rc=$?"
tail -1 gpurun_out/${TAG}_bench.log > gpurun_out/${TAG}_bench.json; wc -c gpurun_out/${TAG}_bench.json
for mode in fp32w fp32 auto; do
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${TAG}_$mode -o bench -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extra --no-product-run --single-stream --precision $mode --extra-out gpurun_out/prof_${TAG}_${mode}_extra.json > gpurun_out/prof_${TAG}_$mode.log 2>&1
  python tools/prof_summary.py gpurun_out/prof_${TAG}_$mode 12 $([ $mode = auto ] && echo --steps-only) > gpurun_out/${TAG}_step_${mode}_kernel_stats.md
  python tools/phase_breakdown.py gpurun_out/prof_${TAG}_$mode 8 > gpurun_out/${TAG}_phases_${mode}.md 2>&1
  find gpurun_out/prof_${TAG}_$mode -name "*kernel_trace.csv" -delete
  head -12 gpurun_out/${TAG}_step_${mode}_kernel_stats.md | cut -c1-160
done
# PMC passes of the dominant conv kernels (counters only, no tracing besides --kernel-trace): tools/pmc_r3.py
i=0
for ctr in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d gpurun_out/pmc_${TAG}/p$i -o p -- python tools/pmc_r3.py > gpurun_out/pmc_${TAG}_p$i.log 2>&1
done
find gpurun_out/pmc_${TAG} -name "*kernel_trace.csv" -delete
python tools/pmc_r3.py --summarise gpurun_out/pmc_${TAG} gpurun_out/${TAG}_conv_pmc | tail -16 | cut -c1-220
# round 6: the F(2,3) split-bf16 kernel against the direct kernels, and the schedule A/B of the auto step (the critical chain alone)
timeout 300 python tools/bench_wino16.py > gpurun_out/${TAG}_wino16_bench.txt 2>&1
timeout 600 python tools/ab_tail.py --precision auto --rounds 2 --only "baseline,chain only (static un-shifted batch: NOT training),mid stage behind R,baseline again" > gpurun_out/${TAG}_ab_schedule.txt 2>&1
timeout 300 python tools/bench_upfused.py f16 f16x2 bf16x3 > gpurun_out/${TAG}_upfused_bench.txt 2>&1
if [ "$3" = "lite" ]; then exit 0; fi
timeout 900 python tools/host_enqueue_n.py 8 auto | tail -1 > gpurun_out/${TAG}_host_enqueue_8.json
# round 5: weight-gradient kernels (micro-benchmark + PMC passes), the reported-only rows of the precision table
timeout 600 python tools/bench_wgrad_direct.py ksweep > gpurun_out/${TAG}_wgrad_bench.txt 2>&1
i=0
for ctr in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_MFMA SQ_INSTS_VMEM_RD" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d gpurun_out/pmcw_${TAG}/p$i -o p -- python tools/pmc_wgrad16.py > gpurun_out/pmcw_${TAG}_p$i.log 2>&1
done
find gpurun_out/pmcw_${TAG} -name "*kernel_trace.csv" -delete
python tools/pmc_wgrad16.py --summarise gpurun_out/pmcw_${TAG} > gpurun_out/${TAG}_wgrad_pmc.txt 2>&1
WGS_FULL_SCHEMES=1 timeout 1500 python -m pytest tests/test_precision_schemes_gpu.py -q -x -k per_scheme > gpurun_out/${TAG}_schemes.log 2>&1; tail -2 gpurun_out/${TAG}_schemes.log
cp gpurun_out/precision_schemes.json gpurun_out/${TAG}_precision_schemes.json 2>/dev/null
