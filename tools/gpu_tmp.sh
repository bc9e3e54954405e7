#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_precision_schemes_gpu.py tests/test_full_size_steps_gpu.py tests/test_proggan_gpu.py tests/test_configs_gpu.py tests/test_precision_policy_cpu.py -q -m gpu -x 2>&1 | tail -4
timeout 600 python tools/ab_tail.py --config cfg2 --precision auto,bf16x3 --steps 12 --warmup 3 --rounds 2 --only "tail default/16" 2>&1 | grep variant
