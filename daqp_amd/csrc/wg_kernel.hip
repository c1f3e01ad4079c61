// wg_kernel.hip -- translation unit of the workgroup-per-problem solve kernel (wg_kernel.hip.h, wg_ldp.hip.h)
#include <hip/hip_runtime.h>
#include "wg_kernel.hip.h"

namespace daqp_amd {
template __global__ void k_ldp_wg<2, false>(BatchDev, int);
template __global__ void k_ldp_wg<2, true>(BatchDev, int);
template __global__ void k_ldp_wg<4, false>(BatchDev, int);
template __global__ void k_ldp_wg<4, true>(BatchDev, int);
template __global__ void k_ldp_wg<4, false, true>(BatchDev, int);
template __global__ void k_ldp_wg<2, false, true>(BatchDev, int);
}
