// tiny_kernel.hip -- translation unit of the 16-problems-per-wave solve kernel for tiny shapes (tiny_kernel.hip.h, tiny_ldp.hip.h):
// generic rows (TRI = 0) and the all-simple-bounds-first shape of config C3 (TRI = 3: ms >= 12), each in the reference's
// arithmetic (FM = false) and with fused multiply-adds (FM = true, the default mode)
#include <hip/hip_runtime.h>
#include "tiny_kernel.hip.h"
#include "tiny_setup.hip.h"

namespace daqp_amd {
template __global__ void k_ldp_tiny<4, 0, false>(const BatchDev *__restrict__, int);
template __global__ void k_ldp_tiny<4, 0, true>(const BatchDev *__restrict__, int);
template __global__ void k_ldp_tiny<4, 3, false>(const BatchDev *__restrict__, int);
template __global__ void k_ldp_tiny<4, 3, true>(const BatchDev *__restrict__, int);
template __global__ void k_setup_tiny<4>(BatchDev, int);
}
