#!/bin/bash
# development build: fewer template instantiations (fast compile). Usage: tools/devbuild.sh [extra hipcc flags]
cd "$(dirname "$0")/../daqp_amd/csrc" || exit 1
mkdir -p ../lib
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -DDAQP_AMD_FEW_VARIANTS "$@" daqp_amd.hip -o ../lib/libdaqp_amd.so 2>&1 | grep -E "error|Function Name|VGPRs:|AGPRs:|Scratch|Spill" | sed 's/\[-Rpass[^]]*\]//g; s/remark: //g; s#^[./a-z_]*.hip.h:[0-9]*:[0-9]*:##'
