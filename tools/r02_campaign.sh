#!/bin/bash
# parity campaigns on the final code of round 2 (exact and default arithmetic)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02g
timeout 2400 python tools/parity_campaign.py 100000 200000 300 > gpurun_out/r02g/parity_campaign.txt 2>&1
timeout 1200 python tools/large_shapes.py > gpurun_out/r02g/large_shapes.txt 2>&1
timeout 1200 python tools/prox_campaign.py > gpurun_out/r02g/prox_campaign.txt 2>&1
