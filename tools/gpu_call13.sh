#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_proggan_gpu.py tests/test_full_size_steps_gpu.py -q -m gpu -x 2>&1 | tail -3
timeout 600 python tools/ab_tail.py --config cfg2 --precision auto --steps 12 --warmup 3 --rounds 2 --only "tail off,tail default/16,tail default/32,tail 256/16" 2>&1 | grep variant
timeout 600 python tools/ab_tail.py --config cfg5 --precision auto --steps 20 --warmup 4 --rounds 2 2>&1 | grep variant
timeout 600 python tools/ab_tail.py --config cfg3 --precision auto --steps 50 --rounds 2 --only "tail off,tail default/16" 2>&1 | grep variant
