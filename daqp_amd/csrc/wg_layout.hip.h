// wg_layout.hip.h -- what the host needs to know about the workgroup solve kernel (wg_kernel.hip.h): its wave limit and the
// size of its dynamic LDS.  Kept apart so that the kernel itself compiles in its own translation unit.
#pragma once

namespace daqp_amd {

constexpr int kWgMaxWaves = 8;    // 512 threads: two waves per SIMD, i.e. 256 VGPRs each (the master's chains need them: at 168 they spilled)

__host__ __device__ inline int wg_round_up(int a, int b) { return (a + b - 1) / b * b; }
// double offset of packed L, and the bytes of dynamic LDS for capL rows of it
__host__ __device__ inline int wg_lds_L(int C, int m) { const int CAP = 64 * C; return 8 * CAP + 2 * 258 + 136 * kWgMaxWaves + 24 + wg_round_up(5 * CAP + 16 + wg_round_up(m, 4), 4) / 2; }
// row stride of the workgroup kernel's active-row scratch: whole 32-column chunks (the default mode's Gram pass reads a row as
// ldr / 32 loads of 16 lanes x 16 bytes; the pad columns are zero)
__host__ __device__ inline int wg_row_stride(int n) { return wg_round_up(n, 32); }
__host__ __device__ inline int wg_lds_bytes(int C, int m, int capL) { return 8 * (wg_lds_L(C, m) + wg_round_up(capL * (capL + 1) / 2, 2)); }

} // namespace daqp_amd
