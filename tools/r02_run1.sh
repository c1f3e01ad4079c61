#!/bin/bash
# round 2, GPU run 1: full GPU test suite, the bench line with every config, rocprof of C4
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02a
python - > gpurun_out/r02a/gen_c4.log 2>&1 <<'PY'
import time, torch
from daqp_amd.synthetic import generate_batch_torch
t0 = time.time(); q = generate_batch_torch(2048, 200, 600, 0, 80, seed=1); torch.cuda.synchronize(); print("C4 gen 2048:", time.time() - t0, "s")
PY
timeout 2400 python -m pytest tests -m gpu -q --maxfail=30 > gpurun_out/r02a/pytest.log 2>&1
echo "pytest exit $?" >> gpurun_out/r02a/pytest.log
timeout 1200 python bench.py > gpurun_out/r02a/bench.json 2> gpurun_out/r02a/bench.err
echo "bench exit $?" >> gpurun_out/r02a/bench.err
