// cycles per v_mfma_f64_16x16x4_f64 on gfx950 (4 independent accumulators, back to back), one wave per SIMD
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4d __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(64) void k(double *out, double *sink, int steps)
{
    const int lane = threadIdx.x;
    double a = 1.0 + lane * 1e-3, b = 1.0 - lane * 1e-3;
    v4d c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    long long t0 = __builtin_readcyclecounter();
    for (int s = 0; s < steps; ++s) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
        }
    }
    long long t1 = __builtin_readcyclecounter();
    if (lane == 0 && blockIdx.x == 0) out[0] = (double)(t1 - t0) / (steps * 16.0);
    sink[blockIdx.x * 64 + lane] = c0[0] + c1[1] + c2[2] + c3[3];
}
int main() {
    double *out, *sink; hipMalloc(&out, 64); hipMalloc(&sink, 4096 * 64 * 8);
    for (int grid : {1, 1024, 4096}) {
        hipLaunchKernelGGL(k, dim3(grid), dim3(64), 0, 0, out, sink, 200);
        hipDeviceSynchronize();
        double h; hipMemcpy(&h, out, 8, hipMemcpyDeviceToHost);
        printf("grid %d: %.1f cycles per v_mfma_f64_16x16x4_f64 (1024 FMA each)\n", grid, h);
    }
    return 0;
}
