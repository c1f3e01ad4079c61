#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02c
for g in 96 128 192 256; do
  echo "== DAQP_AMD_WG_GRID=$g" >> gpurun_out/r02c/c4_grid.log
  DAQP_AMD_WG_GRID=$g timeout 600 python tools/c4_rate.py 4096 prof 2>&1 | grep -v amdgpu.ids >> gpurun_out/r02c/c4_grid.log
done
