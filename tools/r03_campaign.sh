#!/bin/bash
# parity campaigns on the final code of round 3 (exact and default arithmetic)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03g
timeout 2400 python tools/parity_campaign.py 100000 200000 300 > gpurun_out/r03g/parity_campaign.txt 2>&1
timeout 1200 python tools/large_shapes.py > gpurun_out/r03g/large_shapes.txt 2>&1
timeout 1200 python tools/prox_campaign.py > gpurun_out/r03g/prox_campaign.txt 2>&1
