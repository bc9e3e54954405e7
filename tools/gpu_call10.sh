#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_stem_s2d_gpu.py tests/test_reconstructor_gpu.py tests/test_conv_gpu.py tests/test_lenet_gpu.py tests/test_cfg1_step_gpu.py -q -m gpu -x 2>&1 | tail -8
timeout 300 python tools/bench_wgrad_stem.py 2>&1 | tail -12
