#!/bin/bash
# SQ counter passes of the C3 solve / setup launches (round 3: the 16-problems-per-wave kernel).  Usage: tools/r03_c3pmc.sh <tag>
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=${1:-r03}
O=gpurun_out/$TAG; rm -rf $O; mkdir -p $O
CMD="python bench.py --config C3 --steps 2 --warmup 1 --cpu-sample 0 --side-configs none"
i=0; dirs=""
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM" \
           "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM" $EXTRA; do
  i=$((i+1)); rm -rf /tmp/pmc_$i
  timeout 600 rocprofv3 --pmc $grp --output-format csv -d /tmp/pmc_$i -o p -- $CMD > /tmp/pmc_$i.log 2>&1
  echo "pass $i ($grp): exit $?" >> $O/pmc_passes.txt
  grep '^{' /tmp/pmc_$i.log | cut -c1-300 >> $O/pmc_passes.txt
  dirs="$dirs /tmp/pmc_$i"
done
python tools/pmc_json.py $O/pmc_raw_C3.json $dirs > $O/pmc_print_C3.txt 2>&1
python - <<PY
import json
d=json.load(open("$O/pmc_raw_C3.json"))
for k,v in d.items():
    if "k_ldp" in k or "k_setup" in k or "tiny" in k:
        c={n:x["mean"] for n,x in v.items()}
        wc=c.get("SQ_WAVE_CYCLES",0) or 1
        print(k[:60]); print("  ", {n: round(x) for n,x in c.items()})
        print("   active_any %.3f wait_any %.3f wait_inst_any %.3f valu_active %.3f lds_active %.3f" % tuple(c.get(n,0)/wc for n in ("SQ_ACTIVE_INST_ANY","SQ_WAIT_ANY","SQ_WAIT_INST_ANY","SQ_ACTIVE_INST_VALU","SQ_ACTIVE_INST_LDS")))
PY
