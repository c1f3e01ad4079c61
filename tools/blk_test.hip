// tools/blk_test.hip -- blk_factor (csrc/setup_blk.hip.h) on its own: random SPD matrices, one wavefront each, R^-1 against a plain
// host Cholesky + inverse.  hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Idaqp_amd/csrc tools/blk_test.hip -o tools/blk_test.bin
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "setup_blk.hip.h"
using namespace daqp_amd;

template <int NT>
__global__ __launch_bounds__(64) void k_test(const double *H, int n, double *Xout, double *piv, long long *cyc)
{
    __shared__ __attribute__((aligned(16))) double Hs[64 * 64 > blk_scratch_doubles<4>() ? 64 * 64 : blk_scratch_doubles<4>()];
    const int q = blockIdx.x, lane = threadIdx.x;
    for (int e = lane; e < n * n; e += 64) Hs[e] = H[(size_t)q * n * n + e];
    WSYNC();
    blk_v4d X[NT * (NT + 1) / 2];
    double pmin = 1e300, pmax = 0;
    const long long t0 = __builtin_readcyclecounter();
    blk_v4d T[NT * (NT + 1) / 2];
    int offd;
    blk_load<NT>(Hs, n, 1e-11, T, offd);
    double pmn = 1e300, pmx = 0; const bool ok = blk_factor<NT>(T, n, 1e-11, X, Hs, pmn, pmx) && __any(offd); pmin = pmn; pmax = pmx;
    const long long t1 = __builtin_readcyclecounter();
    const int lr = lane & 15, lk = lane >> 4;
    static_for<NT>([&](auto Ic) {
        static_for<NT>([&](auto Jc) {
            constexpr int I = Ic, J = Jc;
            if constexpr (I <= J) {
                for (int r = 0; r < 4; ++r) {
                    const int i = 16 * I + lk + 4 * r, j = 16 * J + lr;
                    if (i < n && j < n) Xout[(size_t)q * n * n + i * n + j] = X[blk_tix<NT>(I, J)][r];
                }
            }
        });
    });
    if (lane == 0) { piv[3 * q] = pmin; piv[3 * q + 1] = pmax; piv[3 * q + 2] = ok ? 1 : 0; cyc[q] = t1 - t0; }
}

int main(int argc, char **argv)
{
    const int n = argc > 1 ? atoi(argv[1]) : 50, N = argc > 2 ? atoi(argv[2]) : 4096;
    const int NT = (n + 15) / 16;
    std::vector<double> H((size_t)N * n * n), X((size_t)N * n * n, 0.0), piv(3 * N);
    srand(1);
    for (int q = 0; q < N; ++q) {
        std::vector<double> G(n * n);
        for (auto &g : G) g = rand() / (double)RAND_MAX - 0.5;
        for (int i = 0; i < n; ++i)
            for (int j = 0; j < n; ++j) {
                double s = (i == j) ? 0.5 : 0.0;
                for (int k = 0; k < n; ++k) s += G[i * n + k] * G[j * n + k];
                H[(size_t)q * n * n + i * n + j] = s * (1.0 + 1e-3 * (i < j ? 1 : (i > j ? -1 : 0)));   // slightly unsymmetric: 1/2 (H + H') matters
            }
    }
    double *dH, *dX, *dp; long long *dc;
    hipMalloc(&dH, H.size() * 8); hipMalloc(&dX, X.size() * 8); hipMalloc(&dp, piv.size() * 8); hipMalloc(&dc, N * 8);
    hipMemcpy(dH, H.data(), H.size() * 8, hipMemcpyHostToDevice);
    hipMemset(dX, 0, X.size() * 8);
    for (int rep = 0; rep < 2; ++rep) {
        if (NT == 1) hipLaunchKernelGGL(k_test<1>, dim3(N), dim3(64), 0, 0, dH, n, dX, dp, dc);
        if (NT == 2) hipLaunchKernelGGL(k_test<2>, dim3(N), dim3(64), 0, 0, dH, n, dX, dp, dc);
        if (NT == 3) hipLaunchKernelGGL(k_test<3>, dim3(N), dim3(64), 0, 0, dH, n, dX, dp, dc);
        if (NT == 4) hipLaunchKernelGGL(k_test<4>, dim3(N), dim3(64), 0, 0, dH, n, dX, dp, dc);
    }
    if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 1; }
    std::vector<long long> cyc(N);
    hipMemcpy(X.data(), dX, X.size() * 8, hipMemcpyDeviceToHost);
    hipMemcpy(piv.data(), dp, piv.size() * 8, hipMemcpyDeviceToHost);
    hipMemcpy(cyc.data(), dc, N * 8, hipMemcpyDeviceToHost);
    double worst = 0, worst_p = 0;
    long long csum = 0;
    for (int q = 0; q < N; ++q) {
        csum += cyc[q];
        if (q >= 64) continue;
        // host: upper Cholesky of 1/2 (H + H'), then the inverse
        std::vector<double> R(n * n, 0.0), Xi(n * n, 0.0);
        const double *h = &H[(size_t)q * n * n];
        double pmin = 1e300, pmax = 0;
        for (int i = 0; i < n; ++i) {
            for (int j = i; j < n; ++j) {
                double s = (i == j) ? h[i * n + i] : 0.5 * (h[i * n + j] + h[j * n + i]);
                for (int k = 0; k < i; ++k) s -= R[k * n + i] * R[k * n + j];
                if (i == j) { pmin = fmin(pmin, s); pmax = fmax(pmax, s); R[i * n + i] = sqrt(s); }
                else R[i * n + j] = s / R[i * n + i];
            }
        }
        for (int j = 0; j < n; ++j) {
            Xi[j * n + j] = 1 / R[j * n + j];
            for (int i = j - 1; i >= 0; --i) {
                double s = 0;
                for (int k = i + 1; k <= j; ++k) s += R[i * n + k] * Xi[k * n + j];
                Xi[i * n + j] = -s / R[i * n + i];
            }
        }
        double scale = 0;
        for (auto v : Xi) scale = fmax(scale, fabs(v));
        for (int e = 0; e < n * n; ++e) worst = fmax(worst, fabs(X[(size_t)q * n * n + e] - Xi[e]) / scale);
        worst_p = fmax(worst_p, fmax(fabs(piv[3 * q] - pmin) / pmin, fabs(piv[3 * q + 1] - pmax) / pmax));
        if (piv[3 * q + 2] != 1) printf("problem %d: not ok\n", q);
    }
    printf("n %d NT %d: max |X - X_host| / max|X| = %.3e, pivots rel %.3e, cycles per problem %.0f\n", n, NT, worst, worst_p, (double)csum / N);
    return 0;
}
