#!/bin/bash
# round 4: feasibility-scan variants of the workgroup kernel (tools/variants.sh wg_kernel.hip <tag>:"<flags>" ...), C4 phase probes
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r04s}; mkdir -p $O; shift
for v in base "$@"; do
  lib=""; [ "$v" != "base" ] && lib=$GRAFT_REPO_ROOT/daqp_amd/lib/variants/libdaqp_amd_$v.so
  echo "== variant $v" >> $O/scan_variants.txt
  DAQP_AMD_LIBRARY=$lib timeout 600 python tools/c4_rate.py 4096 prof 2>&1 | grep -v amdgpu.ids >> $O/scan_variants.txt
done
cat $O/scan_variants.txt
