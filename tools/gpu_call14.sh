#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_biggan_gpu.py tests/test_proggan_gpu.py tests/test_stylegan2_gpu.py tests/test_train_step_gpu.py tests/test_configs_gpu.py -q -m gpu -x 2>&1 | tail -4
timeout 600 python tools/ab_tail.py --config cfg4 --precision auto --steps 60 --warmup 6 --rounds 2 --only "tail off,tail default/16,tail default/32" 2>&1 | grep variant
