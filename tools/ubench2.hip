// micro-costs of the fused Cholesky/inverse update on gfx950 (cycles per update), variants of instruction order
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <utility>
template <class F, int... Is>
__device__ __forceinline__ void sf_impl(F &&f, std::integer_sequence<int, Is...>) { (f(std::integral_constant<int, Is>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void sf(F &&f) { sf_impl(f, std::make_integer_sequence<int, N>{}); }
__device__ __forceinline__ double rl(double v, int src) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_readlane(lo, src); hi = __builtin_amdgcn_readlane(hi, src);
    return __hiloint2double(hi, lo);
}
constexpr int NR = 48;
#define T0 long long t0 = __builtin_readcyclecounter()
#define T1(k) if (threadIdx.x == 0 && blockIdx.x == 0) out[k] = (double)(__builtin_readcyclecounter() - t0) / (steps * NR)
__global__ __launch_bounds__(64) void k(double *out, double *sink, int steps, int kk)
{
    const int lane = threadIdx.x;
    double c[NR + 1], a[NR + 1];
    sf<NR + 1>([&](auto i) __attribute__((always_inline)) { c[i] = 1.0 + lane * 1e-3 + i; a[i] = 0.5 + lane * 1e-4 + i; });
    double rk = 1.0 + lane * 1e-6, col = 1.0 - lane * 1e-6;
    const unsigned sgn = lane > 70 ? 0x80000000u : 0u;
    {   // A: as in the kernel
        T0;
        for (int s = 0; s < steps; ++s) {
            const int k0 = (kk + s) & 7;
            sf<NR>([&](auto ii) __attribute__((always_inline)) {
                constexpr int i = ii;
                const double sc = rl(rk, k0 + 1 + i);
                c[i] = c[i + 1] - sc * rk;
                const double ap = __hiloint2double(__double2hiint(a[i + 1]) | sgn, __double2loint(a[i + 1]));
                a[i] = ap - sc * col;
            });
            rk = c[0] * 1e-9 + 1.0; col = a[0] * 1e-9 + 1.0;
        }
        T1(0);
    }
    {   // B: only the Cholesky half
        T0;
        for (int s = 0; s < steps; ++s) {
            const int k0 = (kk + s) & 7;
            sf<NR>([&](auto ii) __attribute__((always_inline)) {
                constexpr int i = ii;
                const double sc = rl(rk, k0 + 1 + i);
                c[i] = c[i + 1] - sc * rk;
            });
            rk = c[0] * 1e-9 + 1.0;
        }
        T1(1);
    }
    {   // C: groups of 4: readlanes, then 8 muls, then 8 subs
        T0;
        for (int s = 0; s < steps; ++s) {
            const int k0 = (kk + s) & 7;
            sf<NR / 4>([&](auto gg) __attribute__((always_inline)) {
                constexpr int g = gg;
                double sc[4], pc[4], pa[4];
                sf<4>([&](auto h) __attribute__((always_inline)) { sc[h] = rl(rk, k0 + 1 + 4 * g + h); });
                sf<4>([&](auto h) __attribute__((always_inline)) { pc[h] = sc[h] * rk; pa[h] = sc[h] * col; });
                __builtin_amdgcn_sched_barrier(0);
                sf<4>([&](auto h) __attribute__((always_inline)) {
                    constexpr int i = 4 * g + h;
                    c[i] = c[i + 1] - pc[h];
                    const double ap = __hiloint2double(__double2hiint(a[i + 1]) | sgn, __double2loint(a[i + 1]));
                    a[i] = ap - pa[h];
                });
                __builtin_amdgcn_sched_barrier(0);
            });
            rk = c[0] * 1e-9 + 1.0; col = a[0] * 1e-9 + 1.0;
        }
        T1(2);
    }
    {   // D: no readlane at all (scalar from a uniform value): pure VALU cost of 2 mul + 2 sub + or
        T0;
        for (int s = 0; s < steps; ++s) {
            const double sc = 1.0 + s * 1e-9;
            sf<NR>([&](auto ii) __attribute__((always_inline)) {
                constexpr int i = ii;
                c[i] = c[i + 1] - sc * rk;
                const double ap = __hiloint2double(__double2hiint(a[i + 1]) | sgn, __double2loint(a[i + 1]));
                a[i] = ap - sc * col;
            });
            rk = c[0] * 1e-9 + 1.0; col = a[0] * 1e-9 + 1.0;
        }
        T1(3);
    }
    {   // E: readlane with constant lane index (no s_add, immediate lane select)
        T0;
        for (int s = 0; s < steps; ++s) {
            sf<NR>([&](auto ii) __attribute__((always_inline)) {
                constexpr int i = ii;
                const double sc = rl(rk, i + 1);
                c[i] = c[i + 1] - sc * rk;
                const double ap = __hiloint2double(__double2hiint(a[i + 1]) | sgn, __double2loint(a[i + 1]));
                a[i] = ap - sc * col;
            });
            rk = c[0] * 1e-9 + 1.0; col = a[0] * 1e-9 + 1.0;
        }
        T1(4);
    }
    {   // F: fma form (what a relaxed mode would cost): 2 fma + or
        T0;
        for (int s = 0; s < steps; ++s) {
            const int k0 = (kk + s) & 7;
            sf<NR>([&](auto ii) __attribute__((always_inline)) {
                constexpr int i = ii;
                const double sc = rl(rk, k0 + 1 + i);
                c[i] = __builtin_fma(-sc, rk, c[i + 1]);
                a[i] = __builtin_fma(-sc, col, a[i + 1]);
            });
            rk = c[0] * 1e-9 + 1.0; col = a[0] * 1e-9 + 1.0;
        }
        T1(5);
    }
    __shared__ double row[128];
    row[lane] = 1.0 + lane * 1e-6; row[lane + 64] = 1.0;
    __syncthreads();
    {   // G: cholesky half, multiplier broadcast from LDS (row written once per step)
        T0;
        for (int s = 0; s < steps; ++s) {
            const int k0 = (kk + s) & 7;
            row[lane] = rk;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
            const double *rp = row + k0 + 1;
            sf<NR>([&](auto ii) __attribute__((always_inline)) {
                constexpr int i = ii;
                c[i] = c[i + 1] - rp[i] * rk;
            });
            rk = c[0] * 1e-9 + 1.0;
        }
        T1(6);
    }
    {   // H: inverse half, multiplier broadcast from LDS
        T0;
        for (int s = 0; s < steps; ++s) {
            const int k0 = (kk + s) & 7;
            const double *rp = row + k0 + 1;
            sf<NR>([&](auto ii) __attribute__((always_inline)) {
                constexpr int i = ii;
                const double ap = __hiloint2double(__double2hiint(a[i + 1]) | sgn, __double2loint(a[i + 1]));
                a[i] = ap - rp[i] * col;
            });
            col = a[0] * 1e-9 + 1.0;
        }
        T1(7);
    }
    {   // I: fused, multiplier broadcast from LDS
        T0;
        for (int s = 0; s < steps; ++s) {
            const int k0 = (kk + s) & 7;
            row[lane] = rk;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
            const double *rp = row + k0 + 1;
            sf<NR>([&](auto ii) __attribute__((always_inline)) {
                constexpr int i = ii;
                const double sc = rp[i];
                c[i] = c[i + 1] - sc * rk;
                const double ap = __hiloint2double(__double2hiint(a[i + 1]) | sgn, __double2loint(a[i + 1]));
                a[i] = ap - sc * col;
            });
            rk = c[0] * 1e-9 + 1.0; col = a[0] * 1e-9 + 1.0;
        }
        T1(8);
    }
    double acc = rk + col;
    sf<NR + 1>([&](auto i) __attribute__((always_inline)) { acc += c[i] + a[i]; });
    sink[blockIdx.x * 64 + lane] = acc;
}
int main() {
    double *out, *sink; hipMalloc(&out, 64 * 8); hipMalloc(&sink, 1024 * 64 * 8);
    const int steps = 100;
    for (int grid : {1, 1024}) {
        hipLaunchKernelGGL(k, dim3(grid), dim3(64), 40 * 1024, 0, out, sink, steps, 3);
        hipDeviceSynchronize();
        std::vector<double> h(9); hipMemcpy(h.data(), out, 9 * 8, hipMemcpyDeviceToHost);
        const char *nm[] = {"A fused update (as kernel)", "B cholesky half only", "C grouped by 4: rl | mul | sub", "D no readlane", "E const-lane readlane", "F fma form", "G chol half, LDS broadcast", "H inverse half, LDS broadcast", "I fused, LDS broadcast"};
        printf("grid %d:\n", grid);
        for (int i = 0; i < 9; ++i) printf("  %-36s %6.1f cycles per update\n", nm[i], h[i]);
    }
    return 0;
}
