#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for e in "X=1" "WGS_WINO_SMALL=1" "WGS_F32_SMALL=1" "WGS_WINO_SMALL=1 WGS_F32_SMALL=1"; do
  for ss in "" "--single-stream"; do
  env $e timeout 300 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-extra --no-product-run --no-direct-run --no-roofline --precision fp32w $ss --extra-out gpurun_out/c9_x.json 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$e $ss', d['value'], d['ms_per_step'])"
  done
done
