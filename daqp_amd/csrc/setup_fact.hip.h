// setup_fact.hip.h -- k_fact_wg: the Cholesky factor of 1/2 (H + H') and its inverse (utils.c:318-352, 380-389) for the generic setup
// (64 < n <= 200) in the default arithmetic, as its OWN launch in front of k_setup: one WORKGROUP of eight waves per problem, the
// packed upper triangle resident in LDS (157 KB at n = 200), rank-16 updates on the f64 matrix cores.
//
// Why (round 4's probes, profiles/r04c_c4_phases.txt): inside k_setup one wave owns a problem and keeps both factors in HBM scratch
// (2 x 160 KB do not fit LDS); its left-looking Cholesky and its row-by-row inverse keep the reference's operation order and are
// chains of dependent trips to L2 -- 4.7 M + 5.1 M cycles per problem of config C4, 23 of the setup's 39 ms, with the vector pipe a
// fifth busy.  Here a problem has a whole CU: the triangle never leaves LDS between the first load of H and the last store of R^-1,
// a panel of sixteen rows is factored with two workgroup barriers per pivot, and everything behind the panel -- (n - k)^2 / 2 entries
// per sixteen pivots -- is one pass of v_mfma_f64_16x16x4 tiles shared by the eight waves.  The inverse runs block row by block row
// from the bottom: X_IJ = -X_II sum_{I < K <= J} R_IK X_KJ, sixteen-row tiles on the matrix cores again, in place.
//
// Same mathematics as the reference, different summation order: R^-1 agrees to ~1e-15 relative (the exact mode, the regularising
// passes, diagonal Hessians and anything this kernel gives up on -- a pivot at or below zero_tol -- keep k_setup's own, ordered code).
// k_setup reads b.fact[q] = {done, smallest pivot, largest pivot}: done = 1 -> it skips its own two phases and takes R^-1 from the
// scratch (packed, b.setup_g) and the zero-padded square image (b.setup_sq) that this kernel wrote.
#pragma once
#include "batch_dev.hip.h"

namespace daqp_amd {

constexpr int kFactMaxN = 200;        // packed triangle + one 16 x 16 tile of workspace (+ 32 doubles: 288 >= n for v) within 160 KB of LDS
__host__ __device__ inline size_t fact_lds_bytes(int n) { return ((size_t)round_up(n * (n + 1) / 2, 2) + 256 + 32) * 8; }

__global__ void k_fact_wg(BatchDev b);      // defined once, in setup_kernel.hip (DAQP_AMD_SETUP_FACT_IMPL)

#ifdef DAQP_AMD_SETUP_FACT_IMPL
// inverse of a 16 x 16 upper-triangular block of the packed triangle (1 / u_ii on its diagonal), ib <= 16 rows valid, into T[16][16]
// (zero below the diagonal and beyond ib): lane <-> column, the column in registers, U as broadcast reads.  One wave.
__device__ __forceinline__ void fact_tri_inv16(const double *R, int i0, int ib, int n, double *T, int lane)
{
    const int j = lane & 15;
    double x[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) x[i] = 0.0;
#pragma unroll
    for (int i = 15; i >= 0; --i) {                  // column j, bottom up: x_i = -(1 / u_ii) sum_{k > i} U[i][k] x_k (x_k = 0 beyond j)
        if (i < ib) {
            const int pi = roff(i0 + i, n) + i0;
            double sacc = 0;
#pragma unroll
            for (int k = i + 1; k < 16; ++k) if (k < ib) sacc = __builtin_fma(R[pi + k], x[k], sacc);
            const double dinv = R[pi + i];
            x[i] = (j == i) ? dinv : -dinv * sacc;
        }
    }
    if (lane < 16) {
#pragma unroll
        for (int i = 0; i < 16; ++i) T[i * 16 + j] = (j < ib) ? x[i] : 0.0;
    }
}
__global__ __launch_bounds__(512) void k_fact_wg(BatchDev b)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    typedef double v4d __attribute__((ext_vector_type(4)));
    const int q = blockIdx.x;
    const int tid = (int)threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = b.n, rtri = b.rtri;
    double *R = smem, *T = smem + round_up(rtri, 2);
    double *rec = b.fact + (size_t)q * 4;
    const double *H = b.H + (size_t)q * n * n;
    const double zt = b.st.zero_tol;
    const int lr = lane & 15, lk = lane >> 4;
    // optional phase cycle counters -> b.prof[q][22..24]: load + symmetrise, Cholesky, inverse + stores
    long long ft0 = b.prof ? (long long)__builtin_readcyclecounter() : 0;
#define FPROF(slot) do { if (b.prof) { const long long t1_ = (long long)__builtin_readcyclecounter(); if (tid == 0) b.prof[(size_t)q * 32 + 22 + (slot)] = t1_ - ft0; ft0 = t1_; } } while (0)

    // ---- 1/2 (H + H') into the packed triangle; a diagonal H is k_setup's RinvD branch (utils.c:245-312)
    // (two passes over the rows of H, both coalesced and sixteen loads deep: the upper part of row i lands in its own places, then --
    //  behind a barrier -- the lower part of row i is added to column i of the triangle)
    int offd = 0;
    for (int i0 = wv; i0 < n; i0 += 32) {                    // four rows per wave and trip, a row as four masked 64-column pieces: sixteen loads in flight
        double h[4][4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = i0 + 8 * r;
#pragma unroll
            for (int c = 0; c < 4; ++c) { const int j = lane + 64 * c; h[r][c] = (i < n && j < n) ? H[(size_t)i * n + j] : 0.0; }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = i0 + 8 * r;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int j = lane + 64 * c;
                if (i < n && j < n && j >= i) {
                    if (j > i && (h[r][c] > zt || h[r][c] < -zt)) offd = 1;
                    R[roff(i, n) + j] = (j > i) ? 0.5 * h[r][c] : h[r][c];
                }
            }
        }
    }
    __syncthreads();
    for (int i0 = wv; i0 < n; i0 += 32) {                    // the lower part of row i, halved, onto column i of the triangle
        double h[4][4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = i0 + 8 * r;
#pragma unroll
            for (int c = 0; c < 4; ++c) { const int j = lane + 64 * c; h[r][c] = (i < n && j < i) ? H[(size_t)i * n + j] : 0.0; }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = i0 + 8 * r;
#pragma unroll
            for (int c = 0; c < 4; ++c) { const int j = lane + 64 * c; if (i < n && j < i) R[roff(j, n) + i] += 0.5 * h[r][c]; }
        }
    }
    if (!__syncthreads_or(offd)) { if (tid == 0) rec[0] = 0.0; return; }
    FPROF(0);

    // ---- Cholesky, upper, 1 / r_kk on the diagonal.  Panels of sixteen rows, three workgroup barriers each:
    //   (a) the panel's 16 x 16 diagonal block by sixteen lanes of wave 0, lane <-> column, the column in registers, row k's entries
    //       of the other columns through v_readlane with constant lanes;
    //   (b) the rest of the panel, R12 = U11^-T A12, one thread per column, sixteen entries in registers, U11 as broadcast reads;
    //   (c) behind the panel, C_IJ -= R12_I' R12_J tile by tile on the matrix cores, all waves.
    double pmin = DAQP_INF, pmax = 0.0;
    bool fail = false;
    double *piv = T + 256;                                   // [0..15] the panel's pivots, [16] != 0: one of them failed
    for (int K = 0; K < n; K += 16) {
        const int Kend = (K + 16 < n) ? K + 16 : n, kb = Kend - K;
        if (wv == 0) {
            const int j = lane & 15;
            double c[16];
#pragma unroll
            for (int p = 0; p < 16; ++p) c[p] = (p <= j && j < kb) ? R[roff(K + p, n) + K + j] : 0.0;
            double bad = 0.0;
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                if (k < kb) {
                    const double d = rl(c[k], k);
                    if (!(d > zt)) bad = 1.0;
                    if (lane == 0) piv[k] = d;
                    const double inv = rsqrt(d);
                    c[k] = (j == k) ? inv : c[k] * inv;
#pragma unroll
                    for (int i = k + 1; i < 16; ++i) c[i] = __builtin_fma(-rl(c[k], i), c[k], c[i]);
                }
            }
            if (lane < 16) {
#pragma unroll
                for (int p = 0; p < 16; ++p) if (p <= j && j < kb) R[roff(K + p, n) + K + j] = c[p];
            }
            if (lane == 0) piv[16] = bad;
            if (Kend < n) fact_tri_inv16(R, K, 16, n, T, lane);      // V = U11^-1 for (b) (this wave's own LDS stores above are in order before these reads)
        }
        __syncthreads();
        if (piv[16] != 0.0) { fail = true; break; }         // (the same word in every thread: a uniform exit; NaN pivots included)
        for (int k = 0; k < kb; ++k) { const double d = piv[k]; if (d < pmin) pmin = d; if (d > pmax) pmax = d; }
        if (Kend >= n) break;
        const int nbt = (n - Kend + 15) >> 4;
        for (int t = wv; t < nbt; t += 8) {                  // (b) R12 = V' A12, a sixteen-column tile per wave and trip, in place
            const int jc = Kend + 16 * t + lr;
            v4d acc = (v4d){0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const double a = T[(4 * s + lk) * 16 + lr];                                  // V'[i][k] = V[k][i]
                const double bb = (jc < n) ? R[roff(K + 4 * s + lk, n) + jc] : 0.0;
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bb, acc, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) if (jc < n) R[roff(K + lk + 4 * r, n) + jc] = acc[r];
        }
        __syncthreads();
        const int ntiles = nbt * (nbt + 1) / 2;
        for (int t = wv; t < ntiles; t += 8) {
            int I = 0, rem = t;
            while (rem >= nbt - I) { rem -= nbt - I; ++I; }
            const int i0 = Kend + 16 * I, j0 = Kend + 16 * (I + rem);
            const int jc = j0 + lr;
            v4d acc;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = i0 + lk + 4 * r;
                acc[r] = (i < n && jc < n && jc >= i) ? R[roff(i, n) + jc] : 0.0;
            }
#pragma unroll
            for (int s = 0; s < 4; ++s) {                   // (a panel with rows behind it is a full one: sixteen pivots)
                const int pk = roff(K + 4 * s + lk, n);
                const double a = (i0 + lr < n) ? -R[pk + i0 + lr] : 0.0;
                const double bb = (jc < n) ? R[pk + jc] : 0.0;
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bb, acc, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = i0 + lk + 4 * r;
                if (i < n && jc < n && jc >= i) R[roff(i, n) + jc] = acc[r];
            }
        }
        __syncthreads();
    }
    if (fail) { if (tid == 0) rec[0] = 0.0; return; }       // (a uniform decision: k_setup's ordered code takes the problem and reports)
    FPROF(1);

    // ---- R -> R^-1 in place, block rows of sixteen from the bottom.  X_II (sixteen columns, one thread each) goes through T, zero below
    // its diagonal; every tile of the block row is formed in registers by all waves BEFORE any of them is stored (tile (I, J) reads
    // R_IK for I < K <= J: the places of the tiles left of it)
    const int nb = (n + 15) >> 4;
    for (int I = nb - 1; I >= 0; --I) {
        const int i0 = 16 * I, ib = (n - i0 < 16) ? n - i0 : 16;
        if (wv == 0) fact_tri_inv16(R, i0, ib, n, T, lane);
        __syncthreads();
        // tiles right of the diagonal block: tile t costs t + 1 block products -- wave w takes t = w and, beyond eight tiles, its
        // mirror nt - 1 - w (n <= 200: nb <= 13, nt <= 12), so that every wave's pair costs nt + 1
        const int nt = nb - 1 - I;
        v4d out0 = (v4d){0.0, 0.0, 0.0, 0.0}, out1 = (v4d){0.0, 0.0, 0.0, 0.0};
        auto tile_of = [&](int u) __attribute__((always_inline)) {
            if (nt <= 8) return (u == 0 && wv < nt) ? wv : -1;
            if (u == 0) return wv < (nt + 1) / 2 ? wv : -1;
            return (wv < nt / 2) ? nt - 1 - wv : -1;
        };
        for (int u = 0; u < 2; ++u) {
            const int t = tile_of(u);
            if (t < 0) continue;
            const int J = I + 1 + t, j0 = 16 * J, jc = j0 + lr;
            v4d s4 = (v4d){0.0, 0.0, 0.0, 0.0};
            const int ia = i0 + lr;
            const int pia = roff(ia < n ? ia : 0, n);
            for (int K = I + 1; K <= J; ++K) {
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const int kk = 16 * K + 4 * s + lk;
                    const double a = (ia < n && kk < n) ? R[pia + kk] : 0.0;                 // R_IK[i][k]
                    const double bb = (kk < n && jc < n && jc >= kk) ? R[roff(kk < n ? kk : 0, n) + jc] : 0.0;   // X_KJ[k][j], zero below the diagonal of X_JJ
                    s4 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bb, s4, 0, 0, 0);
                }
            }
            v4d o4 = (v4d){0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int s = 0; s < 4; ++s)                      // (the accumulator's register s IS the B operand of k step s: rows 4 s + (l >> 4))
                o4 = __builtin_amdgcn_mfma_f64_16x16x4f64(-T[lr * 16 + 4 * s + lk], s4[s], o4, 0, 0, 0);
            if (u == 0) out0 = o4; else out1 = o4;
        }
        __syncthreads();
        for (int u = 0; u < 2; ++u) {
            const int t = tile_of(u);
            if (t < 0) continue;
            const int jc = 16 * (I + 1 + t) + lr;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = i0 + lk + 4 * r;
                if (i < n && jc < n) R[roff(i, n) + jc] = (u == 0) ? out0[r] : out1[r];
            }
        }
        if (tid < 256) {
            const int i = tid >> 4, j = tid & 15;
            if (i < ib && j < ib && j >= i) R[roff(i0 + i, n) + i0 + j] = T[i * 16 + j];
        }
        __syncthreads();
    }

    // ---- out: packed (what k_setup and the solve read) and as the zero-padded square image (the matrix cores' B operand in k_setup_m)
    {
        double *Ro = b.setup_g + (size_t)q * 2 * (size_t)round_up(rtri, 2) + round_up(rtri, 2);
        for (int e = tid; e < rtri; e += 512) Ro[e] = R[e];
        const int sq_ld = round_up(n, 16);
        double *Rsq = b.setup_sq + (size_t)q * round_up(n, 32) * sq_ld;
        for (int k = wv; k < n; k += 8) {
            const int pk = roff(k, n);
            for (int j = k + lane; j < n; j += 64) Rsq[(size_t)k * sq_ld + j] = R[pk + j];
        }
        // v = R^-T f and x_unc = -R^-1 v (utils.c:474-497, 618-662) while R^-1 is still in LDS: inside k_setup a lane walked a column of the
        // packed factor in HBM term by term -- 1.25 M cycles of a wave per problem for two mat-vecs.  One thread per entry; v goes through
        // the workspace tile (free now), both land in b.v / b.xunc, where k_setup picks them up (it decides whether x_unc is used).
        const double *f = b.f + (size_t)q * n;
        __syncthreads();
        if (tid < n) T[tid] = f[tid];                        // (f through the tile first, then v in its place)
        __syncthreads();
        double vi = 0;
        if (tid < n) {
            const int i = tid;
            double a0 = 0, a1 = 0;
            int j = 0;
            for (; j + 1 <= i; j += 2) { a0 = __builtin_fma(R[roff(j, n) + i], T[j], a0); a1 = __builtin_fma(R[roff(j + 1, n) + i], T[j + 1], a1); }
            if (j <= i) a0 = __builtin_fma(R[roff(j, n) + i], T[j], a0);
            vi = a0 + a1;
            b.v[(size_t)q * n + i] = vi;
        }
        __syncthreads();
        if (tid < n) T[tid] = vi;
        __syncthreads();
        if (tid < n) {
            const int i = tid, pi = roff(i, n);
            double a0 = 0, a1 = 0;
            int j = i;
            for (; j + 1 < n; j += 2) { a0 = __builtin_fma(R[pi + j], T[j], a0); a1 = __builtin_fma(R[pi + j + 1], T[j + 1], a1); }
            if (j < n) a0 = __builtin_fma(R[pi + j], T[j], a0);
            b.xunc[(size_t)q * n + i] = -(a0 + a1);
        }
        if (tid == 0) { rec[1] = pmin; rec[2] = pmax; rec[0] = 1.0; }
    }
    FPROF(2);
#undef FPROF
}
#endif // DAQP_AMD_SETUP_FACT_IMPL

} // namespace daqp_amd
