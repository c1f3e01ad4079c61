#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02d
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fast_mode.py tests/test_gpu_reference_cases.py -m gpu -q -x > gpurun_out/r02d/pytest.log 2>&1
echo "pytest exit $?" >> gpurun_out/r02d/pytest.log
timeout 600 python tools/config_sweep.py 1.0 C3 > gpurun_out/r02d/c3.log 2>&1
