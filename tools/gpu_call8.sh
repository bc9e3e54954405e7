#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_training_trajectory_gpu.py tests/test_train_step_gpu.py tests/test_cfg1_step_gpu.py -q -m gpu -x 2>&1 | tail -3
timeout 900 python tools/ab_tail.py --rounds 3 > gpurun_out/c8_ab.log 2>&1; echo "ab rc=$?"
grep variant gpurun_out/c8_ab.log
