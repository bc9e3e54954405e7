#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_train_step_gpu.py tests/test_training_trajectory_gpu.py tests/test_configs_gpu.py tests/test_full_size_steps_gpu.py -q -m gpu -x 2>&1 | tail -4
timeout 600 python tools/ab_tail.py --config cfg3 --precision fp32w,auto --steps 60 --rounds 3 --only "weights inline,weights ahead" 2>&1 | grep variant
