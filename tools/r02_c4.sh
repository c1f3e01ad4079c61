#!/bin/bash
# C4 rate of the workgroup kernel with phase counters (N = 2048) and at full size
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02c
timeout 600 python tools/c4_rate.py 2048 prof > gpurun_out/r02c/c4_wg.log 2>&1
timeout 600 python tools/c4_rate.py 10000 > gpurun_out/r02c/c4_wg_full.log 2>&1
