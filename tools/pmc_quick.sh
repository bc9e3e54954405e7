#!/bin/bash
# usage: bash tools/pmc_quick.sh <tag> <label filter> — the five PMC passes of tools/pmc_r3.py restricted to the shapes whose label contains the
# filter (e.g. "bf16x3"), summarised into gpurun_out/<tag>_pmc_table.md / .json.  Counters only (no tracing besides --kernel-trace).
TAG=$1; export PMC_ONLY="$2"
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
i=0
for ctr in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d gpurun_out/pmc_${TAG}/p$i -o p -- python tools/pmc_r3.py > gpurun_out/pmc_${TAG}_p$i.log 2>&1
done
find gpurun_out/pmc_${TAG} -name "*kernel_trace.csv" -delete
python tools/pmc_r3.py --summarise gpurun_out/pmc_${TAG} gpurun_out/${TAG}_pmc | cut -c1-260
