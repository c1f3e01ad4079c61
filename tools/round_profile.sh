#!/bin/bash
# round profiles (usage: tools/round_profile.sh <tag>): kernel-trace stats of the bench command (all configs), then PMC passes per config
# (own run per counter group, --pmc only: no tracing domains), summarised as profiles/<tag>_pmc_summary.json
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r03p}; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -o kt -- python bench.py --steps 5 --warmup 1 --cpu-sample 0 --full-out $O/bench_under_trace_full.json > $O/bench_under_trace.json 2> $O/trace.err
find /tmp/kt -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
# the headline alone (no side configs, no batch sweep, no exact-mode leg): here a kernel's average IS the C2 launch
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt2 -o kt -- python bench.py --config C2 --steps 10 --warmup 2 --cpu-sample 0 --side-configs none --no-sweep --no-exact --full-out $O/bench_headline_under_trace_full.json > $O/bench_headline_under_trace.json 2>> $O/trace.err
find /tmp/kt2 -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_headline.csv \;
for cfg in C2 C3 C4 C5; do
  CMD="python bench.py --config $cfg --steps 2 --warmup 1 --cpu-sample 0 --side-configs none --full-out /tmp/pmc_full.json"
  i=0
  dirs=""
  for grp in "FETCH_SIZE" "WRITE_SIZE" \
             "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM" \
             "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM"; do
    i=$((i+1)); rm -rf /tmp/pmc_${cfg}_$i
    timeout 900 rocprofv3 --pmc $grp --output-format csv -d /tmp/pmc_${cfg}_$i -o p -- $CMD > /tmp/pmc_${cfg}_$i.log 2>&1
    echo "$cfg pass $i ($grp): exit $?" >> $O/pmc_passes.txt
    grep '^{' /tmp/pmc_${cfg}_$i.log | cut -c1-200 >> $O/pmc_passes.txt
    dirs="$dirs /tmp/pmc_${cfg}_$i"
  done
  python tools/pmc_json.py $O/pmc_raw_$cfg.json $dirs > $O/pmc_print_$cfg.txt 2>&1
done
python tools/pmc_config_summary.py $O/pmc_summary.json $O/pmc_raw_C2.json $O/pmc_raw_C3.json $O/pmc_raw_C4.json $O/pmc_raw_C5.json $O/kernel_stats.csv > $O/pmc_summary_print.txt 2>&1
timeout 900 python bench.py --full-out $O/bench_full.json > $O/bench.json 2> $O/bench.err
tools/ubench4.bin > $O/ubench4.txt 2>&1
tools/ubench5.bin > $O/ubench5.txt 2>&1
tools/ubench5.bin 512 200 4 >> $O/ubench5.txt 2>&1      # the residency of the tiered launch: two four-wave workgroups per CU
python tools/gpu_profile.py 20000 > $O/c2_phases.txt 2>&1
python tools/c4_rate.py 4096 prof > $O/c4_phases.txt 2>&1
python tools/c5_phases.py > $O/c5_phases.txt 2>&1
python tools/latency_one.py > $O/latency_one.txt 2>&1
timeout 900 python tools/shape_map.py > $O/shape_map.txt 2>&1
