#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for q in 0 3 2; do
  WGS_SIDE_CU_QUARTERS=$q timeout 400 python tools/ab_tail.py --only "tail off,tail 128/16" --steps 50 >> gpurun_out/c4_ab_cu.log 2>&1; echo "q=$q rc=$?"
done
grep variant gpurun_out/c4_ab_cu.log
