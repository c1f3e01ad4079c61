#!/bin/bash
# What a SECOND resident problem per CU is worth to the workgroup kernel, measured where it can be measured without a tiered
# factor: C4's shape (n=200, m=600) with fewer rows active at the optimum, so that the packed factor of every working set fits HALF the LDS
# (DAQP_AMD_WG_CAPL=100: 78 KB per workgroup).  Same problems, same kernel, three residencies:
#   8 waves, one workgroup per CU (the shipped configuration) | 4 waves, one per CU | 4 waves, two per CU (grid 512)
# usage: bash tools/c4_two_per_cu.sh [N] [n_active ...]
N=${1:-4096}; shift
NAS=${@:-"40 30"}
cd "$(dirname "$0")/.."
for na in $NAS; do
  echo "# n_active at the optimum = $na"
  for cfg in "8 0 256" "8 100 256" "4 100 256" "4 100 512"; do
    set -- $cfg
    envs="DAQP_AMD_WG_WAVES=$1 DAQP_AMD_WG_GRID=$3"
    [ "$2" != 0 ] && envs="$envs DAQP_AMD_WG_CAPL=$2"
    echo -n "waves $1, capL ${2/#0/default}, workgroups in flight $3: "
    env C4_NA=$na $envs timeout 600 python tools/c4_rate.py $N 2>/dev/null | tail -1
  done
done
