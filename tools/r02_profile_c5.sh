#!/bin/bash
# PMC passes of one config again (default C5) into gpurun_out/r02p, then the summary over all configs
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02p; mkdir -p $O
for cfg in ${1:-C5}; do
  CMD="python bench.py --config $cfg --steps 2 --warmup 1 --cpu-sample 0 --side-configs none"
  i=0; dirs=""
  for grp in "FETCH_SIZE" "WRITE_SIZE" \
             "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM" \
             "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM"; do
    i=$((i+1)); rm -rf /tmp/pmc_${cfg}_$i
    timeout 900 rocprofv3 --pmc $grp --output-format csv -d /tmp/pmc_${cfg}_$i -o p -- $CMD > /tmp/pmc_${cfg}_$i.log 2>&1
    dirs="$dirs /tmp/pmc_${cfg}_$i"
  done
  python tools/pmc_json.py $O/pmc_raw_$cfg.json $dirs > $O/pmc_print_$cfg.txt 2>&1
done
python tools/pmc_config_summary.py $O/pmc_summary.json $O/pmc_raw_C2.json $O/pmc_raw_C3.json $O/pmc_raw_C4.json $O/pmc_raw_C5.json > $O/pmc_summary_print.txt 2>&1
