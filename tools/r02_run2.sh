#!/bin/bash
# round 2: workgroup kernel bring-up: parity tests of the large shapes, then C4 rates (wg vs one-wave) with phase counters
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02b
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fast_mode.py tests/test_gpu_golden.py -m gpu -q -x -k "workgroup or c4 or C4 or time_limit or large or degenerate_branches" > gpurun_out/r02b/pytest.log 2>&1
echo "pytest exit $?" >> gpurun_out/r02b/pytest.log
timeout 600 python tools/c4_rate.py 2048 prof > gpurun_out/r02b/c4_wg.log 2>&1
DAQP_AMD_NO_WG=1 timeout 600 python tools/c4_rate.py 2048 prof > gpurun_out/r02b/c4_onewave.log 2>&1
timeout 600 python tools/c4_rate.py 10000 > gpurun_out/r02b/c4_wg_full.log 2>&1
