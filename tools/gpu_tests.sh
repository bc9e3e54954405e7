#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -q -m gpu -x > gpurun_out/full_tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/full_tests.log
tail -8 gpurun_out/full_tests.log
