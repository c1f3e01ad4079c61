#!/bin/bash
# the opt-in 16-problems-per-wavefront SOLVE kernel of tiny shapes is not in the default library: this links
# daqp_amd/lib/variants/libdaqp_amd_tiny.so = the current full build + tiny_kernel.hip, host code compiled with -DDAQP_AMD_WITH_TINY.
#   tools/tinybuild.sh && DAQP_AMD_LIBRARY=$PWD/daqp_amd/lib/variants/libdaqp_amd_tiny.so DAQP_AMD_TINY=1 python -m pytest tests/test_gpu_tiny.py -m gpu
cd "$(dirname "$0")/.." || exit 1
python -c "import daqp_amd; daqp_amd.build()" || exit 1
V=daqp_amd/lib/variants; mkdir -p $V
F="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -DDAQP_AMD_WITH_TINY"
hipcc $F -c daqp_amd/csrc/daqp_amd.hip -o $V/daqp_amd.hip.tiny.o &
hipcc $F -c daqp_amd/csrc/tiny_kernel.hip -o $V/tiny_kernel.hip.tiny.o &
wait
OBJS=""
for o in daqp_amd/lib/obj/*.hip.o; do case "$(basename $o)" in daqp_amd.hip.o|tiny_kernel.hip.o) ;; *) OBJS="$OBJS $o";; esac; done
hipcc --offload-arch=gfx950 -fPIC -shared $OBJS $V/daqp_amd.hip.tiny.o $V/tiny_kernel.hip.tiny.o -o $V/libdaqp_amd_tiny.so && echo "built $V/libdaqp_amd_tiny.so"
