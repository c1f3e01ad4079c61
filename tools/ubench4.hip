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

// d = bit(lane) of m ? b : a, the mask an SGPR pair computed ahead of the chain (no v_cmp + VCC round trip per pivot)
__device__ __forceinline__ double selm(unsigned long long m, double b, double a)
{
    int alo = __double2loint(a), ahi = __double2hiint(a), blo = __double2loint(b), bhi = __double2hiint(b), dlo, dhi;
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(dlo) : "v"(alo), "v"(blo), "s"(m));
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(dhi) : "v"(ahi), "v"(bhi), "s"(m));
    return __hiloint2double(dhi, dlo);
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
            } else if (MODE == 8) {   // select replaced by an EXEC mask around one fma
                const double bj = rl(x, 8 + q);
                const unsigned long long m = (1ull << (8 + q)) - 1;
                asm volatile("s_mov_b64 exec, %[m]\n\tv_fma_f64 %[x], -%[b], %[L], %[x]\n\ts_mov_b64 exec, -1" : [x] "+v"(x) : [m] "s"(m), [b] "s"(bj), [L] "v"(L));
            } else if (MODE == 9) {   // the same with separate multiply and add (reference rounding)
                const double bj = rl(x, 8 + q);
                const unsigned long long m = (1ull << (8 + q)) - 1;
                double t;
                asm volatile("s_mov_b64 exec, %[m]\n\tv_mul_f64 %[t], %[b], %[L]\n\tv_add_f64 %[x], %[x], -%[t]\n\ts_mov_b64 exec, -1" : [x] "+v"(x), [t] "=&v"(t) : [m] "s"(m), [b] "s"(bj), [L] "v"(L));
            } else if (MODE == 11) {  // run-time group, select by a precomputed SGPR mask (mul + add)
                const int l0 = (r & 7) * 8;
                const double bj = rl(x, l0 + q);
                const unsigned long long m = ((1ull << l0) << q) - 1;
                x = selm(m, x - bj * L, x);
            } else if (MODE == 12) {  // the same with fma
                const int l0 = (r & 7) * 8;
                const double bj = rl(x, l0 + q);
                const unsigned long long m = ((1ull << l0) << q) - 1;
                x = selm(m, __builtin_fma(-bj, L, x), x);
            } else if (MODE == 13) {  // run-time group, multiplier zeroed ahead of the chain, fma
                const int l0 = (r & 7) * 8;
                const double bj = rl(x, l0 + q);
                const unsigned long long m = ((1ull << l0) << q) - 1;
                x = __builtin_fma(-bj, selm(m, L, 0.0), x);
            } else if (MODE == 10) {  // EXEC mask, run-time group (lane = 8 (r & 7) + q)
                const int l0 = (r & 7) * 8;
                const double bj = rl(x, l0 + q);
                const unsigned long long m = ((1ull << l0) << q) - 1;
                asm volatile("s_mov_b64 exec, %[m]\n\tv_fma_f64 %[x], -%[b], %[L], %[x]\n\ts_mov_b64 exec, -1" : [x] "+v"(x) : [m] "s"(m), [b] "s"(bj), [L] "v"(L));
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
                           "pivot step without select", "pivot through LDS (write, read back uniform)", "two interleaved pivot chains (per pair)", "pivot step with fma", "pivot step: EXEC-masked fma", "pivot step: EXEC-masked mul + add", "pivot step: EXEC-masked fma, run-time group", "run-time group, SGPR-mask select, mul + add", "run-time group, SGPR-mask select, fma", "run-time group, pre-masked multiplier, fma"};
    std::vector<long long> hc(nb);
    for (int mode = 0; mode < 14; ++mode) {
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
            case 8: hipLaunchKernelGGL(k<8>, dim3(nb), dim3(64), 0, 0, out, cyc, in, reps); break;
            case 9: hipLaunchKernelGGL(k<9>, dim3(nb), dim3(64), 0, 0, out, cyc, in, reps); break;
            case 10: hipLaunchKernelGGL(k<10>, dim3(nb), dim3(64), 0, 0, out, cyc, in, reps); break;
            case 11: hipLaunchKernelGGL(k<11>, dim3(nb), dim3(64), 0, 0, out, cyc, in, reps); break;
            case 12: hipLaunchKernelGGL(k<12>, dim3(nb), dim3(64), 0, 0, out, cyc, in, reps); break;
            case 13: hipLaunchKernelGGL(k<13>, dim3(nb), dim3(64), 0, 0, out, cyc, in, reps); break;
            }
            hipDeviceSynchronize();
        }
        hipMemcpy(hc.data(), cyc, nb * 8, hipMemcpyDeviceToHost);
        double s = 0; for (auto c : hc) s += c;
        printf("%-55s %7.1f cycles per step (%d blocks, one wave each)\n", names[mode], s / nb / reps / 8, nb);
    }
    return 0;
}
