// wg_layout.hip.h -- what the host needs to know about the workgroup solve kernel (wg_kernel.hip.h): its wave limit and the
// size of its dynamic LDS.  Kept apart so that the kernel itself compiles in its own translation unit.
#pragma once

namespace daqp_amd {

constexpr int kWgMaxWaves = 8;    // 512 threads: two waves per SIMD, i.e. 256 VGPRs each (the master's chains need them: at 168 they spilled)

__host__ __device__ inline int wg_round_up(int a, int b) { return (a + b - 1) / b * b; }
// double offset of packed L, and the bytes of dynamic LDS for capL rows of it
__host__ __device__ inline int wg_lds_L(int C, int m) { const int CAP = 64 * C; return 8 * CAP + 2 * 258 + 72 * kWgMaxWaves + 20 + wg_round_up(5 * CAP + 16 + wg_round_up(m, 4), 4) / 2; }
__host__ __device__ inline int wg_lds_bytes(int C, int m, int capL) { return 8 * (wg_lds_L(C, m) + wg_round_up(capL * (capL + 1) / 2, 2)); }

} // namespace daqp_amd
