#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_proggan_gpu.py -q -m gpu -x -s -k f16_backward 2>&1 | grep "d/dshift\|passed\|failed\|Error\|assert" | cut -c1-200
