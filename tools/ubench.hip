// micro-costs on gfx950 for the idioms of the one-wave-per-QP kernels (cycles per step, s_memtime)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__device__ __forceinline__ double rl(double v, int src) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_readlane(lo, src); hi = __builtin_amdgcn_readlane(hi, src);
    return __hiloint2double(hi, lo);
}
#define T0 long long t0 = __builtin_readcyclecounter()
#define T1(k) if (threadIdx.x == 0 && blockIdx.x == 0) out[k] = (double)(__builtin_readcyclecounter() - t0) / steps
__global__ __launch_bounds__(64) void k(double *out, double *sink, int steps, int n)
{
    extern __shared__ double lds[];
    const int lane = threadIdx.x;
    for (int i = lane; i < 4096; i += 64) lds[i] = 1.0 + i * 1e-3;
    __syncthreads();
    double acc = lane, p = 1.0 + lane * 1e-3;
    { T0; for (int s = 0; s < steps; ++s) for (int j = 0; j < n; ++j) acc -= rl(p, j); T1(0); }
    { T0; for (int s = 0; s < steps; ++s) { _Pragma("unroll") for (int j = 0; j < 16; ++j) acc -= rl(p, j); } T1(1); }
    { T0; for (int s = 0; s < steps; ++s) { _Pragma("unroll") for (int j = 0; j < 16; ++j) acc = acc * 1.0000001 - p; } T1(2); }
    { T0; double a0 = acc, a1 = p, a2 = 3, a3 = 4;
      for (int s = 0; s < steps; ++s) { _Pragma("unroll") for (int j = 0; j < 4; ++j) { a0 = a0 * 1.0000001 - p; a1 = a1 * 1.0000001 - p; a2 = a2 * 1.0000001 - p; a3 = a3 * 1.0000001 - p; } }
      acc += a0 + a1 + a2 + a3; T1(3); }
    { T0; int idx = lane; double v = 0; for (int s = 0; s < steps; ++s) { _Pragma("unroll") for (int j = 0; j < 16; ++j) { v = lds[idx & 4095]; idx = (int)v + idx + 1; } } acc += v; T1(4); }
    { T0; double v = 0; for (int s = 0; s < steps; ++s) { double r[16]; _Pragma("unroll") for (int j = 0; j < 16; ++j) r[j] = lds[(lane + 64 * j + s) & 4095]; _Pragma("unroll") for (int j = 0; j < 16; ++j) v += r[j]; } acc += v; T1(5); }
    { T0; double v = 0; for (int s = 0; s < steps; ++s) { double r[16]; _Pragma("unroll") for (int j = 0; j < 16; ++j) r[j] = lds[(j * 8 + s) & 4095]; _Pragma("unroll") for (int j = 0; j < 16; ++j) v += r[j]; } acc += v; T1(6); }
    { T0; for (int s = 0; s < steps; ++s) { _Pragma("unroll") for (int j = 0; j < 16; ++j) lds[(lane * 17 + j + s) & 4095] = acc + j; } T1(7); }
    { T0; for (int s = 0; s < steps; ++s) { if (lane == (s & 63)) { _Pragma("unroll") for (int j = 0; j < 16; ++j) lds[(j + s) & 4095] = acc + j; } } T1(8); }
    { T0; int c = 0; for (int s = 0; s < steps; ++s) { _Pragma("unroll") for (int j = 0; j < 16; ++j) { if (__builtin_amdgcn_readfirstlane(c + j) < n) c += 1; else c += 2; } } acc += c; T1(9); }
    { T0; for (int s = 0; s < steps; ++s) { _Pragma("unroll") for (int j = 0; j < 16; ++j) acc = acc / (p + j); } T1(10); }
    { T0; for (int s = 0; s < steps; ++s) { _Pragma("unroll") for (int j = 0; j < 16; ++j) acc = sqrt(acc * acc + 1.0); } T1(11); }
    sink[blockIdx.x * 64 + lane] = acc;
}
int main() {
    double *out, *sink; hipMalloc(&out, 64 * 8); hipMalloc(&sink, 1024 * 64 * 8);
    const int steps = 200;
    for (int grid : {1, 1024}) {
        hipLaunchKernelGGL(k, dim3(grid), dim3(64), 40 * 1024, 0, out, sink, steps, 16);
        hipDeviceSynchronize();
        std::vector<double> h(12); hipMemcpy(h.data(), out, 12 * 8, hipMemcpyDeviceToHost);
        const char *nm[] = {"rolled readlane chain (16/step)", "unrolled readlane chain x16", "dependent mul+add x16", "4 indep chains x4 (16 mul+add)",
                            "dependent LDS reads x16", "16 indep LDS reads + adds", "16 broadcast LDS reads + adds", "16 LDS writes (all lanes)",
                            "16 LDS writes (one lane)", "16 uniform branches", "16 dependent f64 div", "16 dependent f64 sqrt"};
        printf("grid %d (LDS 40 KB/WG => <= 4 waves/CU):\n", grid);
        for (int i = 0; i < 12; ++i) printf("  %-36s %8.1f cycles per group of 16 => %6.1f each\n", nm[i], h[i], h[i] / 16);
    }
    return 0;
}
