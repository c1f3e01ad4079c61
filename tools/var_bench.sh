#!/bin/bash
# bench the default library and the tuning variants built by tools/variants.sh side by side: [SIDE=C3,C4,C5] tools/var_bench.sh <tag> ...
for t in "" "$@"; do
  if [ -n "$t" ]; then export DAQP_AMD_LIBRARY=$GRAFT_REPO_ROOT/daqp_amd/lib/variants/libdaqp_amd_$t.so; else unset DAQP_AMD_LIBRARY; fi
  echo "variant ${t:-default}"
  python bench.py --steps 20 --warmup 2 --side-configs ${SIDE:-C3,C5} --no-exact --cpu-sample 0 --full-out /tmp/var_full.json > /dev/null; python -c "
import json,sys
d=json.load(open('/tmp/var_full.json'))
def sh(t,c):
    r=c['roofline']; print('  ',t, round(c['value']), r.get('pipeline',{}).get('setup_ms'), round(r['avg_launch_ms'],3))
sh('C2',d)
for k,c in d['configs'].items(): sh(k,c)"
done
