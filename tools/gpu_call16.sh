#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_sample_step_gpu.py tests/test_train_step_gpu.py tests/test_training_trajectory_gpu.py tests/test_distributed_gpu.py tests/test_cli_gpu.py -q -m gpu -x -s 2>&1 | grep -v "^$" | tail -12
timeout 600 python tools/ab_tail.py --config cfg3 --precision auto --steps 60 --rounds 3 --only "torch sampler,kernel sampler" 2>&1 | grep variant
