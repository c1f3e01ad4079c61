// tools/ubench4.hip -- latencies behind the master wave's substitution chains (one wave per SIMD):
//   dependent v_add_f64 / v_mul_f64+v_add_f64 chains, the readlane -> mul -> add -> select -> readlane loop of a pivot step,
//   the same with the pivot broadcast through LDS, and DPP row broadcast.
// build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/ubench4.hip -o tools/ubench4.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__device__ __forceinline__ double rl(double v, int src)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_readlane(lo, src);
    hi = __builtin_amdgcn_readlane(hi, src);
    return __hiloint2double(hi, lo);
}

template <int MODE>
__global__ void k(double *out, long long *cyc, const double *in, int reps)
{
    __shared__ double sm[256];
    const int lane = threadIdx.x;
    double x = in[lane], y = in[64 + lane], L = in[128 + lane];
    sm[lane] = y;
    __syncthreads();
    const long long t0 = __builtin_readcyclecounter();
    for (int r = 0; r < reps; ++r) {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            if (MODE == 0) x = x + y;                                           // dependent add
            else if (MODE == 1) x = x * L + y;                                   // dependent mul + add (no contraction)
            else if (MODE == 2) { const double bj = rl(x, 8 + q); const double t = x - bj * L; x = (lane < 8 + q) ? t : x; }   // pivot step, immediate lane
            else if (MODE == 3) { const double bj = rl(x, (r & 7) * 8 + q); const double t = x - bj * L; x = (lane < (r & 7) * 8 + q) ? t : x; }   // runtime lane
            else if (MODE == 4) { const double bj = rl(x, 8 + q); x = x - bj * L; }   // no select
            else if (MODE == 5) {   // pivot through LDS: owner writes, everybody reads (uniform address)
                if (lane == 8 + q) sm[128] = x;
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
                const double bj = sm[128];
                x = x - bj * L;
            } else if (MODE == 6) {   // two independent chains interleaved (ILP check)
                const double bj = rl(x, 8 + q); x = x - bj * L;
                const double cj = rl(y, 8 + q); y = y - cj * L;
            } else if (MODE == 7) {   // fused multiply-add
                const double bj = rl(x, 8 + q); x = __builtin_fma(-bj, L, x);
            }
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 64 + lane] = x + y;
    if (lane == 0) cyc[blockIdx.x] = t1 - t0;
}

int main()
{
    const int reps = 2000, nb = 256 * 4;
    double *out, *in; long long *cyc;
    hipMalloc(&out, nb * 64 * 8); hipMalloc(&in, 192 * 8); hipMalloc(&cyc, nb * 8);
    std::vector<double> h(192);
    for (int i = 0; i < 192; ++i) h[i] = 1.0 + 1e-9 * i;
    for (int i = 128; i < 192; ++i) h[i] = 1e-9 * (i - 100);
    hipMemcpy(in, h.data(), 192 * 8, hipMemcpyHostToDevice);
    const char *names[] = {"dependent v_add_f64", "dependent v_mul_f64 + v_add_f64", "pivot step: readlane(imm) mul add select", "pivot step: readlane(runtime lane) mul add select",
                           "pivot step without select", "pivot through LDS (write, read back uniform)", "two interleaved pivot chains (per pair)", "pivot step with fma"};
    std::vector<long long> hc(nb);
    for (int mode = 0; mode < 8; ++mode) {
        for (int it = 0; it < 2; ++it) {
            switch (mode) {
            case 0: hipLaunchKernelGGL(k<0>, dim3(nb), dim3(64), 0, 0, out, cyc, in, reps); break;
            case 1: hipLaunchKernelGGL(k<1>, dim3(nb), dim3(64), 0, 0, out, cyc, in, reps); break;
            case 2: hipLaunchKernelGGL(k<2>, dim3(nb), dim3(64), 0, 0, out, cyc, in, reps); break;
            case 3: hipLaunchKernelGGL(k<3>, dim3(nb), dim3(64), 0, 0, out, cyc, in, reps); break;
            case 4: hipLaunchKernelGGL(k<4>, dim3(nb), dim3(64), 0, 0, out, cyc, in, reps); break;
            case 5: hipLaunchKernelGGL(k<5>, dim3(nb), dim3(64), 0, 0, out, cyc, in, reps); break;
            case 6: hipLaunchKernelGGL(k<6>, dim3(nb), dim3(64), 0, 0, out, cyc, in, reps); break;
            case 7: hipLaunchKernelGGL(k<7>, dim3(nb), dim3(64), 0, 0, out, cyc, in, reps); break;
            }
            hipDeviceSynchronize();
        }
        hipMemcpy(hc.data(), cyc, nb * 8, hipMemcpyDeviceToHost);
        double s = 0; for (auto c : hc) s += c;
        printf("%-55s %7.1f cycles per step (%d blocks, one wave each)\n", names[mode], s / nb / reps / 8, nb);
    }
    return 0;
}
