#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02e; mkdir -p $O
CMD="python bench.py --config C3 --steps 3 --warmup 1 --cpu-sample 0 --side-configs none"
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt3 -o kt -- $CMD > $O/bench_c3.json 2> $O/trace.err
find /tmp/kt3 -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_c3.csv \;
i=0
for grp in "FETCH_SIZE" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_IFETCH SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM"; do
  i=$((i+1)); rm -rf /tmp/pmc3_$i
  timeout 600 rocprofv3 --pmc $grp --output-format csv -d /tmp/pmc3_$i -o p -- $CMD > /tmp/pmc3_$i.log 2>&1
  echo "pass $i ($grp): exit $?" >> $O/pmc_passes.txt
done
python tools/pmc_json.py $O/pmc_c3.json /tmp/pmc3_1 /tmp/pmc3_2 /tmp/pmc3_3 > $O/pmc_c3_print.txt 2>&1
