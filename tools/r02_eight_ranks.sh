#!/bin/bash
# the driver's 8-rank launch shape on the one GPU of this box (all ranks on cuda:0, gloo rendezvous): north_star's
# "1 M C3 QPs over 8 GPUs" as ONE batch split k mod 8, and the weak-scaling C2 headline with 8 x 12 500 QPs
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02h
L="python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29711"
timeout 1200 $L bench.py --gpus 8 --config C3 --strong --steps 5 --warmup 1 --single-device --backend gloo --cpu-sample 0 > gpurun_out/r02h/c3_strong_8ranks_one_device.json 2> gpurun_out/r02h/c3.err
echo "exit $?" >> gpurun_out/r02h/c3.err
timeout 1200 $L bench.py --gpus 8 --batch 12500 --steps 5 --warmup 1 --single-device --backend gloo --cpu-sample 0 > gpurun_out/r02h/c2_weak_8ranks_one_device.json 2> gpurun_out/r02h/c2.err
echo "exit $?" >> gpurun_out/r02h/c2.err
