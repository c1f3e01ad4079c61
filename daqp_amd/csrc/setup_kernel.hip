// setup_kernel.hip -- translation unit of the one-wave QP -> LDP kernel for n <= 64 (setup_fast.hip.h): every size class in the
// reference's arithmetic, with fused multiply-adds in the Cholesky / inverse sweep (the default mode), and as the regularising
// re-run of flagged problems (PROX: always the reference's arithmetic)
#include <hip/hip_runtime.h>
#include "setup_fast.hip.h"
#include "tiny_setup.hip.h"
#include "setup_blk.hip.h"
#define DAQP_AMD_SETUP_M_IMPL   // (k_setup_m is not a template: defined in this translation unit, declared where it is launched)
#include "setup_m.hip.h"
#define DAQP_AMD_SETUP_FACT_IMPL   // (likewise k_fact_wg)
#include "setup_fact.hip.h"

namespace daqp_amd {
#define DAQP_SETUP_SIZE(NMAX) \
    template __global__ void k_setup_fast<NMAX, false, false>(BatchDev, int); \
    template __global__ void k_setup_fast<NMAX, false, true>(BatchDev, int); \
    template __global__ void k_setup_fast<NMAX, true, false>(BatchDev, int);
DAQP_SETUP_SIZE(16)
DAQP_SETUP_SIZE(32)
DAQP_SETUP_SIZE(56)
DAQP_SETUP_SIZE(64)
#undef DAQP_SETUP_SIZE
// tiny shapes (n <= 12, m <= 48): sixteen problems per wavefront (tiny_setup.hip.h), the default setup for them
template __global__ void k_setup_tiny<4>(BatchDev, int);
// 16 < n <= 64 without simple bounds, default arithmetic: the factorisation on the matrix cores (setup_blk.hip.h)
#define DAQP_BLK_SHAPE(NT, NW, TAIL) template __global__ void k_setup_blk<NT, NW, TAIL>(const BatchDev *__restrict__, int);
DAQP_BLK_SHAPES
#undef DAQP_BLK_SHAPE
}
