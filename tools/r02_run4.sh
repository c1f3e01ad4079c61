#!/bin/bash
# workgroup-kernel / generic-setup work: exact + default mode tests of the large shapes, C4 rate with phase counters
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02d
timeout 1500 python -m pytest tests/test_gpu_fast_mode.py tests/test_gpu_parity.py tests/test_gpu_prox.py -m gpu -q -x -k "fast or c4 or C4 or workgroup or shapes or large or generic or ldp_setup or diagonal or prox" > gpurun_out/r02d/pytest.log 2>&1
echo "pytest exit $?" >> gpurun_out/r02d/pytest.log
timeout 600 python tools/c4_rate.py 2048 prof > gpurun_out/r02d/c4_prof.log 2>&1
timeout 600 python tools/c4_rate.py 10000 > gpurun_out/r02d/c4_full.log 2>&1
timeout 900 python tools/large_shapes.py > gpurun_out/r02d/large_shapes.log 2>&1
