#!/bin/bash
# round 2 profiles: kernel-trace stats of the bench command (all configs), then PMC passes (own run each, no tracing domains)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02p; mkdir -p $O
rocprofv3 -L 2>/dev/null | grep -oE "\b(SQ_[A-Z_0-9]+|FETCH_SIZE|WRITE_SIZE|TCC_[A-Z_0-9]+|GRBM_[A-Z_]+)\b" | sort -u > $O/counters_available.txt
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -o kt -- python bench.py --steps 5 --warmup 1 --cpu-sample 0 > $O/bench_under_trace.json 2> $O/trace.err
find /tmp/kt -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
CMD="python bench.py --steps 2 --warmup 1 --cpu-sample 0 --side-configs C4"
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" \
           "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM" \
           "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM"; do
  i=$((i+1)); rm -rf /tmp/pmc_$i
  timeout 900 rocprofv3 --pmc $grp --output-format csv -d /tmp/pmc_$i -o p -- $CMD > /tmp/pmc_$i.log 2>&1
  echo "pass $i ($grp): exit $?" >> $O/pmc_passes.txt
  tail -3 /tmp/pmc_$i.log >> $O/pmc_passes.txt
done
python tools/pmc_json.py $O/pmc_raw.json /tmp/pmc_1 /tmp/pmc_2 /tmp/pmc_3 /tmp/pmc_4 > $O/pmc_print.txt 2>&1
