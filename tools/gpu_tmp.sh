#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_conv_gpu.py tests/test_reconstructor_gpu.py -q -m gpu -x 2>&1 | tail -2
echo pipelined; timeout 300 python tools/bench_wgrad16.py 2>&1 | tail -14
echo plain; WGS_WGRAD_PLAIN=1 timeout 300 python tools/bench_wgrad16.py 2>&1 | tail -14
