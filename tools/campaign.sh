#!/bin/bash
# parity campaigns on the current code (usage: tools/campaign.sh <tag>: results in gpurun_out/<tag>/) (exact and default arithmetic; default mode without the second pass of recheck.hip.h)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r04p}; mkdir -p $O
timeout 2400 python tools/parity_campaign.py 100000 200000 300 60 > $O/parity_campaign.txt 2>&1
timeout 1200 python tools/large_shapes.py > $O/large_shapes.txt 2>&1
timeout 1200 python tools/prox_campaign.py > $O/prox_campaign.txt 2>&1
# BASELINE.json's full sizes on SURVEY 8(d)'s own draws (C2 100 000, C3 1 000 000, C4 10 000, C5 100 000 x 10 warm steps) against the
# reference library on the host threads: default mode vs the release build, exact mode vs the strict build (bit for bit)
OMP_NUM_THREADS=1 OPENBLAS_NUM_THREADS=1 timeout 900 python tools/full_size_parity.py C2,C3,C4,C5 1 > $O/full_size_parity_default.txt 2>&1
OMP_NUM_THREADS=1 OPENBLAS_NUM_THREADS=1 timeout 900 python tools/full_size_parity.py C2,C3,C4,C5 1 exact > $O/full_size_parity_exact.txt 2>&1
for f in parity_campaign large_shapes prox_campaign; do tail -n 2 $O/$f.txt; done
grep -h "^C[2345]:" $O/full_size_parity_*.txt
