#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02f
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/lt -o lat -- python tools/latency_split.py > gpurun_out/r02f/log.txt 2>&1
find /tmp/lt -name "*kernel_stats.csv" -exec cp {} gpurun_out/r02f/kernel_stats.csv \;
cut -c1-150 gpurun_out/r02f/kernel_stats.csv | head -14
