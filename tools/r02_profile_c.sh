#!/bin/bash
# third (final) profile set of round 2: kernel-trace stats + bench line + C4 phase probes after the setup kernels' occupancy changes
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02r; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -o kt -- python bench.py --steps 5 --warmup 1 --cpu-sample 0 > $O/bench_under_trace.json 2> $O/trace.err
find /tmp/kt -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err
timeout 600 python tools/c4_rate.py 2048 prof > $O/c4_phases.txt 2>&1
