// tiny_kernel.hip -- translation unit of the 16-problems-per-wave solve kernel for tiny shapes (tiny_kernel.hip.h, tiny_ldp.hip.h):
// generic rows (TRI = 0) and the all-simple-bounds-first shape of config C3 (TRI = 3: ms >= 12), each in the reference's
// arithmetic (FM = false) and with fused multiply-adds (FM = true, the default mode).  NOT part of the default build: compiled and linked
// only with -DDAQP_AMD_WITH_TINY (tools/tinybuild.sh; daqp_amd/_lib.py::units) -- the kernel is slower than the register kernel on the
// shape it was written for
#include <hip/hip_runtime.h>
#include "tiny_kernel.hip.h"

namespace daqp_amd {
template __global__ void k_ldp_tiny<4, 0, false>(const BatchDev *__restrict__, int);
template __global__ void k_ldp_tiny<4, 0, true>(const BatchDev *__restrict__, int);
template __global__ void k_ldp_tiny<4, 3, false>(const BatchDev *__restrict__, int);
template __global__ void k_ldp_tiny<4, 3, true>(const BatchDev *__restrict__, int);
}
