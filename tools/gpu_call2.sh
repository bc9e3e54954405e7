#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python tools/ab_tail.py > gpurun_out/c7_ab_tail.log 2>&1; echo "ab rc=$?"
grep variant gpurun_out/c7_ab_tail.log
