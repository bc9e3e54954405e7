#!/bin/bash
# Round profiles on the GPU box: default bench line, PMC passes of the dominant conv launches, rocprofv3 kernel table.
# usage (from the repo root): gpurun -- 'bash tools/run_profiles.sh r2'   -> gpurun_out/<tag>_*; copy the summaries into profiles/
tag=${1:-r2}
mkdir -p gpurun_out/pmc_$tag
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1200 python bench.py > gpurun_out/${tag}_bench.log 2>&1
grep '^{' gpurun_out/${tag}_bench.log | tail -1 > gpurun_out/${tag}_bench_default.json
python - <<PY
import json
d=json.load(open('gpurun_out/${tag}_bench_default.json'))
print(d['value'], d['ms_per_step'], d['config']['precision'], d['roofline']['kernel'], d['roofline']['achieved'], d['roofline']['frac'], d['roofline'].get('traffic'))
print(d['cpu_baseline'])
for e in d['extra']: print({k: e[k] for k in e if k in ('config','precision','value','ms_per_step','error')})
PY
i=0
for ctr in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d gpurun_out/pmc_$tag/p$i -o p -- python tools/pmc_conv16.py f16 > gpurun_out/pmc_$tag/p$i.log 2>&1
done
find gpurun_out/pmc_$tag -name "*kernel_trace.csv" -delete
python tools/pmc_to_json.py gpurun_out/pmc_$tag f16 gpurun_out/${tag}_conv_pmc
WGS_TWO_STREAMS=0 timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -o bench -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extra > gpurun_out/prof_$tag.log 2>&1
python tools/prof_summary.py gpurun_out/prof_$tag 9 > gpurun_out/${tag}_step_kernel_stats.md
head -14 gpurun_out/${tag}_step_kernel_stats.md | cut -c1-150
find gpurun_out/prof_$tag -name "*kernel_trace.csv" -delete
