#!/bin/bash
# parity campaigns on the current code (usage: tools/campaign.sh <tag>: results in gpurun_out/<tag>/) (exact and default arithmetic; default mode without the second pass of recheck.hip.h)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r04p}; mkdir -p $O
timeout 2400 python tools/parity_campaign.py 100000 200000 300 60 > $O/parity_campaign.txt 2>&1
timeout 1200 python tools/large_shapes.py > $O/large_shapes.txt 2>&1
timeout 1200 python tools/prox_campaign.py > $O/prox_campaign.txt 2>&1
for f in parity_campaign large_shapes prox_campaign; do tail -n 2 $O/$f.txt; done
