cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export DAQP_AMD_WG_INVERSE=1
mkdir -p gpurun_out/r02i
timeout 1500 python -m pytest tests/test_gpu_fast_mode.py tests/test_gpu_parity.py tests/test_gpu_golden.py tests/test_gpu_reference_cases.py tests/test_gpu_prox.py -m gpu -q -x -k "fast or c4 or C4 or workgroup or shapes or large or generic or degenerate or golden or shared" > gpurun_out/r02i/pytest.log 2>&1
echo "pytest exit $?" >> gpurun_out/r02i/pytest.log
timeout 900 python tools/large_shapes.py > gpurun_out/r02i/large_shapes.log 2>&1
