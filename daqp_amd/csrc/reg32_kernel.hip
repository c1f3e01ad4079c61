// reg32_kernel.hip -- translation unit of the register-centric solve kernel with an fp32 IMAGE of M in the registers (reg_kernel.hip.h,
// IMG = 1): the shapes whose M itself fills the register file, at two waves per SIMD, in the default arithmetic
#include <hip/hip_runtime.h>
#include "reg_kernel.hip.h"

namespace daqp_amd {
template __global__ void k_ldp_reg<3, 25, true, 2>(const BatchDev *__restrict__, int);    // C2 / C5: n <= 50 with 129 <= m <= 160 (and 33 <= n <= 50 with 65 <= m <= 128)
template __global__ void k_ldp_reg<3, 25, true, 1>(const BatchDev *__restrict__, int);    // n <= 50, 161 <= m <= 192
template __global__ void k_ldp_reg<2, 32, true, 1>(const BatchDev *__restrict__, int);    // 51 <= n <= 63, m <= 128
template __global__ void k_ldp_reg<4, 32, true, 1>(const BatchDev *__restrict__, int);    // n <= 63, m <= 256 where no register shape fits: the image alone (one wave per SIMD), k_ldp behind it
template __global__ void k_ldp_reg<8, 16, true, 1>(const BatchDev *__restrict__, int);    // the same for n <= 32, m <= 512
template __global__ void k_ldp_reg<6, 25, true, 1>(const BatchDev *__restrict__, int);    // n <= 50, m <= 384
template __global__ void k_ldp_reg<5, 32, true, 1>(const BatchDev *__restrict__, int);    // n <= 63, m <= 320
}
