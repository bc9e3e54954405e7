#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 600 python tools/biggan_mixed_sweep.py 128 2>&1 | grep BigGAN
timeout 600 python tools/biggan_mixed_sweep.py 256 2>&1 | grep BigGAN
