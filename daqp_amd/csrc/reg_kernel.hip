// reg_kernel.hip -- translation unit of the register-centric solve kernel (reg_kernel.hip.h, wave_ldp_reg.hip.h): every
// register shape in the reference's arithmetic (FM = false) and with fused multiply-adds (FM = true, the default mode)
#include <hip/hip_runtime.h>
#include "reg_kernel.hip.h"

namespace daqp_amd {
#define DAQP_REG_SHAPE(NB, NP) \
    template __global__ void k_ldp_reg<NB, NP, false>(const BatchDev *__restrict__, int); \
    template __global__ void k_ldp_reg<NB, NP, true>(const BatchDev *__restrict__, int);
DAQP_REG_SHAPE(1, 6)
DAQP_REG_SHAPE(1, 8)
DAQP_REG_SHAPE(3, 25)
#ifndef DAQP_AMD_FEW_VARIANTS
DAQP_REG_SHAPE(1, 13)
DAQP_REG_SHAPE(1, 16)
DAQP_REG_SHAPE(2, 16)
DAQP_REG_SHAPE(3, 8)
DAQP_REG_SHAPE(4, 8)
DAQP_REG_SHAPE(1, 25)
DAQP_REG_SHAPE(2, 32)
#endif
#undef DAQP_REG_SHAPE
}
