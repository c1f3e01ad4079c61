#!/bin/bash
# single-problem path: tests that go through daqp_quadprog / setup_daqp, then the latency tool
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02e
timeout 1500 python -m pytest tests/test_gpu_golden.py tests/test_c_boundary.py tests/test_gpu_reference_cases.py tests/test_gpu_prox.py -m gpu -q -x > gpurun_out/r02e/pytest.log 2>&1
echo "pytest exit $?" >> gpurun_out/r02e/pytest.log
timeout 600 python tools/latency_one.py > gpurun_out/r02e/latency.log 2>&1
DAQP_AMD_NO_POOL=1 timeout 600 python tools/latency_one.py > gpurun_out/r02e/latency_nopool.log 2>&1
