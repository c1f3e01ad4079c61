#!/bin/bash
# usage: tools/pmc.sh "<kernel regex>" <min grid> "<cmd>" "CTR1 CTR2" "CTR3 CTR4" ...
# One rocprofv3 --pmc pass per counter group (own run each, no tracing domains), summarised per dispatch.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
re="$1"; mg="$2"; cmd="$3"; shift 3
i=0
for grp in "$@"; do
  i=$((i+1)); d=/tmp/pmc_$i; rm -rf $d
  rocprofv3 --pmc $grp --output-format csv -d $d -o p -- $cmd > /tmp/pmc_$i.log 2>&1
  python tools/pmc_summary.py $d "$re" $mg
done
