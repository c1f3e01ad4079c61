// setup_blk.hip.h -- the Cholesky factor of 1/2 (H + H') and its inverse (utils.c:318-352, 380-389) for n <= 64 on the f64 matrix cores,
// ONE wavefront per problem, the whole triangle in registers as 16 x 16 tiles (default arithmetic only: same mathematics as the
// reference, another summation order -- R^-1 agrees to ~1e-15 relative; the exact mode keeps setup_fast.hip.h's ordered sweep).
//
// Why: the ordered sweep of setup_fast.hip.h applies every rank-one update with two v_readlane broadcasts per pair of fused
// multiply-adds -- ~30 issue cycles per 128 multiply-adds on at most n of 64 lanes -- and was 110 k of the 265 k cycles a config C2
// problem spends in its setup (tools/gpu_profile.py).  A v_mfma_f64_16x16x4 does 1 024 multiply-adds in 64 cycles with no broadcast
// at all, and in the accumulator layout of that instruction
//       tile element (register r, lane l)  <->  (row (l >> 4) + 4 r, column l & 15)
// a tile is, unchanged, the B operand of a later product (register s = k step s) and, used as the A operand, its own TRANSPOSE.
// The right-looking block Cholesky needs exactly those two: R_KJ = V_K' H_KJ (V_K = R_KK^-1) and H_IJ -= R_KI' R_KJ.  Only the
// sixteen pivots of a diagonal block are serial: that block is gathered lane <-> column (sixteen registers, replicated in the four
// lane groups), factored and inverted with v_readlane broadcasts as before -- 2 x 120 broadcasts per block instead of n^2 per problem.
// The block inverse X = R^-1, X_IJ = -V_I sum_{I < K <= J} R_IK X_KJ, needs R_IK and V_I as UNtransposed A operands: those ten tiles
// take one trip through LDS (2 KB each, all together).  The finished X tiles ARE the B fragments of M = A R^-1 (setup_fast.hip.h's Bf).
#pragma once
#include "wave_ldp_reg.hip.h"
#include "batch_dev.hip.h"
#include "setup_fast.hip.h"

namespace daqp_amd {

typedef double blk_v4d __attribute__((ext_vector_type(4)));
typedef double blk_v2d __attribute__((ext_vector_type(2)));
// (global-memory pointers out of the descriptor: DAQP_GLOBAL / as_global of batch_dev.hip.h -- generic pointers made every access a flat
// operation: the blocked image's stores alternate with LDS reads, 6 k cycles per tile of rows instead of ~2 k)
#define BLK_GLOBAL(T) DAQP_GLOBAL(T)
template <class T> __device__ __forceinline__ auto blk_g(T *p) { return as_global(p); }

// upper-triangular tile grid of NT x NT blocks, row-major over I <= J
template <int NT> __host__ __device__ constexpr int blk_tix(int I, int J) { return I * NT - I * (I - 1) / 2 + (J - I); }
constexpr int kBlkScratch = 16 * 17;   // doubles of LDS one tile transposition takes (odd stride: no bank conflicts)

// lane K of each sixteen-lane row to every lane of that row (the four rows hold the same wherever this is used)
template <int K> __device__ __forceinline__ double blk_bcast(double v) { return __builtin_amdgcn_update_dpp(0.0, v, 0x150 + K, 0xf, 0xf, true); }   // (bound_ctrl: no "old" value to initialise the destination with)
__device__ __forceinline__ double blk_sum16(double v)   // sum over the sixteen lanes of a row group, result in all of them
{
    v += dpp_f64<0xB1>(v);    // quad_perm [1,0,3,2]
    v += dpp_f64<0x4E>(v);    // quad_perm [2,3,0,1]
    v += dpp_f64<0x141>(v);   // row_half_mirror
    v += dpp_f64<0x140>(v);   // row_mirror
    return v;
}

// tiles that the block inverse needs as UNtransposed A operands (register s, lane l: element (l & 15, 4 s + (l >> 4))): the NT (NT - 1) / 2
// off-diagonal tiles of the factor and the inverses of all but the last diagonal block -- one trip through LDS for all of them together
template <int NT> __host__ __device__ constexpr int blk_scratch_doubles() { return (NT * (NT - 1) / 2 + NT - 1) * kBlkScratch; }

// 1/2 (H + H') (utils.c:318-324) from the n x n row-major image of H in LDS into the tiles of the upper triangle (identity in the padding
// rows / columns >= n).  offd: an entry ABOVE the diagonal of H itself exceeds zero_tol (utils.c:245-252 looks at those only).
template <int NT>
__device__ __forceinline__ void blk_load(const double *Hs, int n, double zero_tol, blk_v4d (&T)[NT * (NT + 1) / 2], int &offd)
{
    const int lane = lane_id(), lr = lane & 15, lk = lane >> 4;
    offd = 0;
    static_for<NT>([&](auto Ic) __attribute__((always_inline)) {
        static_for<NT>([&](auto Jc) __attribute__((always_inline)) {
            constexpr int I = Ic, J = Jc;
            if constexpr (I <= J) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int i = 16 * I + lk + 4 * r, j = 16 * J + lr;
                    const bool in = i < n && j < n;
                    const double hij = Hs[in ? i * n + j : 0], hji = Hs[in ? j * n + i : 0];
                    if (in && j > i && (hij > zero_tol || hij < -zero_tol)) offd = 1;
                    T[blk_tix<NT>(I, J)][r] = in ? ((i == j) ? hij : 0.5 * (hij + hji)) : ((i == j) ? 1.0 : 0.0);
                }
            }
        });
    });
}

// Factor and invert.  T: the tiles blk_load left (used up).  X[blk_tix(I, J)]: on return the tiles of R^-1 (zero below the diagonal;
// identity in the padding).  scr: blk_scratch_doubles<NT>() doubles of LDS.  pmin / pmax: smallest / largest pivot (utils.c:354-356).
// Returns false when a pivot is not safely positive (<= 2 zero_tol, or not a number): the caller's ordered code decides then.
template <int NT>
__device__ __forceinline__ bool blk_factor(blk_v4d (&T)[NT * (NT + 1) / 2], int n, double zero_tol, blk_v4d (&X)[NT * (NT + 1) / 2], double *scr, double &pmin, double &pmax)
{
    const int lane = lane_id(), lr = lane & 15, lk = lane >> 4;
    blk_v4d V[NT];                      // inverses of the factor's diagonal blocks
    // the pivots, wave-uniform values in vector registers: smallest / largest of the real ones, and their plain sum (a NaN shows there)
    double dmin = DAQP_INF, dmax = 0.0, dsum = 0.0;
    static_for<NT>([&](auto Kc) __attribute__((always_inline)) {
        constexpr int K = Kc;
        // ---- the diagonal block, lane <-> column (its sixteen rows in registers; the four lane groups hold the same).  Entries below the
        // diagonal are whatever the symmetric tile holds there: every update below only ever reads row k right of column k.
        double c[16];
        {
            const blk_v4d d = T[blk_tix<NT>(K, K)];
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int g = 0; g < 4; ++g) c[4 * r + g] = __shfl(d[r], 16 * g + lr);
        }
        // (broadcasts: v_mov_b64_dpp row_newbcast -- lane k of each sixteen-lane row to the whole row, ONE vector instruction and no scalar
        //  registers, where a v_readlane pair per value kept a hundred scalars in flight per step and spilled them to lanes)
        static_for<16>([&](auto kc) __attribute__((always_inline)) {
            constexpr int k = kc;
            const double d = blk_bcast<k>(c[k]);
            const bool real = 16 * K + k < n;
            dmin = fmin(dmin, real ? d : (double)DAQP_INF);
            dmax = fmax(dmax, real ? d : 0.0);
            dsum += d;
            const double inv = rsqrt(d);
            c[k] = (lr == k) ? inv : c[k] * inv;     // (1 / r_kk on the diagonal, as the reference stores it)
            static_for<15 - k>([&](auto ii) __attribute__((always_inline)) {
                constexpr int i = k + 1 + ii;
                c[i] = __builtin_fma(-blk_bcast<i>(c[k]), c[k], c[i]);
            });
        });
        // V = U^-1, column lr bottom up: x_i = -(1 / u_ii) sum_{k > i} U[i][k] x_k (U[i][k]: row i of the factor, lane k); lanes left of
        // column i hold zeros in every x_k, k > i, so their x_i comes out as -0.0 by itself
        // (the rows pass through an empty asm first: the broadcasts below are the SAME expressions as in the sweep above, and kept for
        //  reuse they are 120 live doubles.  A FUSED sweep -- the inverse's accumulators updated off the sweep's own broadcasts, as in
        //  setup_fast.hip.h -- was built and measured: 39 k instead of 48 k cycles for the factorisation on its own, but with the
        //  accumulators next to the rows the kernel no longer fits 256 registers (181 spilled) and its factorisation phase took 86 k
        //  instead of 62 k cycles.)
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("" : "+v"(c[i]));
        double x[16];
        static_for<16>([&](auto ic) __attribute__((always_inline)) {
            constexpr int i = 15 - ic;
            double sacc = 0;
            static_for<15 - i>([&](auto kk) __attribute__((always_inline)) {
                constexpr int k = i + 1 + kk;
                sacc = __builtin_fma(blk_bcast<k>(c[i]), x[k], sacc);
            });
            const double dinv = blk_bcast<i>(c[i]);
            x[i] = (lr == i) ? dinv : -dinv * sacc;
        });
        {   // into the tile layout: this lane's rows are lk + 4 r.  (The four candidates pass through an empty asm first: as elements of
            //  x[] the selects are turned into ONE load at a selected address -- x[] then lives in scratch.)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                double x0 = x[4 * r], x1 = x[4 * r + 1], x2 = x[4 * r + 2], x3 = x[4 * r + 3];
                asm volatile("" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3));
                double v = x0;
                v = (lk == 1) ? x1 : v;
                v = (lk == 2) ? x2 : v;
                v = (lk == 3) ? x3 : v;
                V[K][r] = v;
            }
        }
        // ---- the block row right of it: R_KJ = V' H_KJ
        static_for<NT>([&](auto Jc) __attribute__((always_inline)) {
            constexpr int J = Jc;
            if constexpr (J > K) {
                blk_v4d acc = (blk_v4d){0.0, 0.0, 0.0, 0.0};
                const blk_v4d h = T[blk_tix<NT>(K, J)];
#pragma unroll
                for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(V[K][s], h[s], acc, 0, 0, 0);
                T[blk_tix<NT>(K, J)] = acc;
            }
        });
        // ---- everything behind it: H_IJ -= R_KI' R_KJ
        static_for<NT>([&](auto Ic) __attribute__((always_inline)) {
            static_for<NT>([&](auto Jc) __attribute__((always_inline)) {
                constexpr int I = Ic, J = Jc;
                if constexpr (I > K && J >= I) {
                    blk_v4d acc = T[blk_tix<NT>(I, J)];
                    const blk_v4d a = T[blk_tix<NT>(K, I)], bb = T[blk_tix<NT>(K, J)];
#pragma unroll
                    for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-a[s], bb[s], acc, 0, 0, 0);
                    T[blk_tix<NT>(I, J)] = acc;
                }
            });
        });
    });
    // ---- the factor's off-diagonal tiles and V_0 .. V_{NT-2} as untransposed A operands (register s, lane l: element (l & 15, 4 s + (l >> 4))):
    // all stores, then all loads
    blk_v4d VT[NT > 1 ? NT - 1 : 1];
    if constexpr (NT > 1) {
        WSYNC();
        static_for<NT>([&](auto Ic) __attribute__((always_inline)) {
            static_for<NT>([&](auto Jc) __attribute__((always_inline)) {
                constexpr int I = Ic, J = Jc;
                if constexpr (I < J) {
                    double *sc = scr + (blk_tix<NT>(I, J) - I - 1) * kBlkScratch;
#pragma unroll
                    for (int r = 0; r < 4; ++r) sc[(lk + 4 * r) * 17 + lr] = T[blk_tix<NT>(I, J)][r];
                }
            });
            if constexpr (Ic < NT - 1) {
                double *sc = scr + (NT * (NT - 1) / 2 + Ic) * kBlkScratch;
#pragma unroll
                for (int r = 0; r < 4; ++r) sc[(lk + 4 * r) * 17 + lr] = V[Ic][r];
            }
        });
        WSYNC();
        static_for<NT>([&](auto Ic) __attribute__((always_inline)) {
            static_for<NT>([&](auto Jc) __attribute__((always_inline)) {
                constexpr int I = Ic, J = Jc;
                if constexpr (I < J) {
                    const double *sc = scr + (blk_tix<NT>(I, J) - I - 1) * kBlkScratch;
#pragma unroll
                    for (int s_ = 0; s_ < 4; ++s_) T[blk_tix<NT>(I, J)][s_] = sc[lr * 17 + 4 * s_ + lk];
                }
            });
            if constexpr (Ic < NT - 1) {
                const double *sc = scr + (NT * (NT - 1) / 2 + Ic) * kBlkScratch;
#pragma unroll
                for (int s_ = 0; s_ < 4; ++s_) VT[Ic][s_] = sc[lr * 17 + 4 * s_ + lk];
            }
        });
    }
    // ---- X = R^-1 by block columns, bottom up inside a column: X_JJ = V_J, X_IJ = -V_I sum_{I < K <= J} R_IK X_KJ
    static_for<NT>([&](auto Jc) __attribute__((always_inline)) {
        constexpr int J = Jc;
        X[blk_tix<NT>(J, J)] = V[J];
        static_for<J>([&](auto Ii) __attribute__((always_inline)) {
            constexpr int I = J - 1 - Ii;
            blk_v4d s4 = (blk_v4d){0.0, 0.0, 0.0, 0.0};
            static_for<J - I>([&](auto Kk) __attribute__((always_inline)) {
                constexpr int K = I + 1 + Kk;
                const blk_v4d at = T[blk_tix<NT>(I, K)], xb = X[blk_tix<NT>(K, J)];
#pragma unroll
                for (int s_ = 0; s_ < 4; ++s_) s4 = __builtin_amdgcn_mfma_f64_16x16x4f64(at[s_], xb[s_], s4, 0, 0, 0);
            });
            blk_v4d o4 = (blk_v4d){0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int s_ = 0; s_ < 4; ++s_) o4 = __builtin_amdgcn_mfma_f64_16x16x4f64(-VT[I][s_], s4[s_], o4, 0, 0, 0);
            X[blk_tix<NT>(I, J)] = o4;
        });
    });
    // the pivots: all safely positive (and numbers)?
    pmin = rl(dmin, 0); pmax = rl(dmax, 0);
    const double ds0 = rl(dsum, 0);
    const bool ok = (ds0 == ds0) && (pmin > 2.0 * zero_tol);
    return ok;
}

// ------------------------------------------------------------------------------------------------------------------------------------
// k_setup_blk<NT>: QP -> LDP for 16 < n <= 16 NT <= 64 without simple bounds (ms == 0), default arithmetic -- k_setup_fast<.., FM>
// with the factorisation above instead of the ordered sweep.  Everything that is not the factorisation keeps that kernel's code and
// order of steps (utils.c:58-687): bound check, v = R^-T f, the unconstrained optimum, M = A R^-1 on the matrix cores with the
// R^-1 tiles as resident B fragments, row normalisation, d, the blocked image.  A diagonal H takes the RinvD branch (utils.c:245-312).
// A problem whose pivots are not safely positive, or whose pivot ratio is within a factor two of the singularity test
// (utils.c:354-356), leaves with DAQP_NEEDS_ORDERED and nothing else written: the launch of k_setup_fast<.., FM> that follows in the
// stream takes exactly those problems (mask bit kSetupOnlyMarked) and decides in the ordered arithmetic, as before this kernel existed.
// ------------------------------------------------------------------------------------------------------------------------------------
// rows of A per LDS tile: a multiple of sixteen (only the last tile of a problem then holds a partial row tile of the matrix cores: 150
// rows are 10 of them, where 56 + 56 + 38 was 4 + 4 + 3), as many as fit where H's image and the factorisation's scratch have been
__host__ __device__ inline int blk_region(int n, int scratch) { const int h = n * n + 8; return h > scratch ? h : scratch; }
__host__ __device__ inline int blk_tile_rows(int n, int scratch)
{
    const int ldr = n | 1;
    int tr = 64;
    while (tr > 16 && tr * ldr > blk_region(n, scratch)) tr -= 16;
    return tr;
}
// LDS: [ H's image, then the factorisation's scratch, then the tiles of A / M ] f (later: the rows' sums of squares) | v (later: the
// rows' products with v) | x_unc | sense
struct BlkLds { int R, fv, vv, xu, sens, total_bytes; };
template <int NT>
__host__ __device__ inline BlkLds blk_lds(int n, int m)
{
    BlkLds s;
    const int ldr = n | 1, TR = blk_tile_rows(n, blk_scratch_doubles<NT>());
    int r0 = blk_region(n, blk_scratch_doubles<NT>());
    if (r0 < TR * ldr) r0 = TR * ldr;
    int o = round_up(r0, 2);
    s.R = 0;
    s.fv = o; o += 64; s.vv = o; o += 64; s.xu = o; o += 64;
    s.sens = o;
    s.total_bytes = o * 8 + round_up(m, 4) * 4;
    return s;
}

// NW: columns of a row of M the kernel handles (n <= NW <= 16 NT, a multiple of 8).
// TAIL: the last column block holds at most four real columns (n - 16 (NT - 1) <= 4: config C2's n = 50 has two).  On the matrix cores
// those columns cost a whole column tile -- 14 of the 38 matrix instructions of a row tile at n = 50 -- so they are formed on the
// vector pipe instead: the A operand a lane already holds (its row, its k's) against the columns' entries broadcast out of the R^-1
// tiles (v_mov_b64_dpp), a sum over the four lane groups, one store into the row.
// The descriptor comes through a pointer (scalar loads at the point of use): as a by-value argument its ~60 pointers are loaded at
// entry and stay live across the factorisation, whose broadcasts need the scalar registers themselves.
template <int NT, int NW = 16 * NT, bool TAIL = false>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_setup_blk(const BatchDev *__restrict__ bp, int mask)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    static_assert(NW <= 16 * NT && NW > 16 * (NT - 1) && NW % 8 == 0, "column width within the last block");
    static_assert(!TAIL || (NT > 1 && NW == 16 * (NT - 1) + 8), "tail columns: the first eight of the last block");
    constexpr int NTT = NT * (NT + 1) / 2, KT = TAIL ? 4 * (NT - 1) + 1 : NW / 4, CT = TAIL ? NT - 1 : NT;   // (k steps of the A operand; column tiles on the matrix cores)
    const BatchDev &b = *bp;
    const int q = blockIdx.x, lane = lane_id(), lr = lane & 15, lk = lane >> 4;
    const int n = b.n, m = b.m, mA = b.mA, ms = b.ms;
    const BlkLds o = blk_lds<NT>(n, m);
    const int TR = blk_tile_rows(n, blk_scratch_doubles<NT>());
    double *Hs = smem + o.R, *fl = smem + o.fv, *vv = smem + o.vv, *xu = smem + o.xu, *tile = smem + o.R;
    double *srow = fl, *drow = vv;      // (f and v have been used up -- v lives on in registers and in HBM -- when the rows' sums arrive)
    const int tailc = TAIL ? n - 16 * (NT - 1) : 0;
    int *sens = reinterpret_cast<int *>(smem + o.sens);
    const double *H = b.H + (size_t)q * n * n, *A = b.A + (size_t)q * mA * n;      // (only ever the source of global -> LDS copies)
    BLK_GLOBAL(QState) *qs = blk_g(b.qs + q);
    const double zero_tol = b.st.zero_tol, primal_tol = b.st.primal_tol;
    const bool force = b.st.eps_prox > 0.0;
    const bool direct = !(n & 1) && (n & 31) && !(((size_t)H | (size_t)A) & 15);   // (as in k_setup_fast: rows copied HBM -> LDS unpadded)
    const int ldr = direct ? n : (n | 1);
    int flag = 1, activate = 0;
    long long pt[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    long long t0 = (kProfile && b.prof) ? (long long)__builtin_readcyclecounter() : 0;
#define SPROF(slot) do { if (kProfile && b.prof) { const long long t1 = (long long)__builtin_readcyclecounter(); pt[slot] += t1 - t0; t0 = t1; } } while (0)

    if (direct) copy_async(Hs, H, n * n);     // in flight while the bounds are checked
    {   // --- sense (utils.c:84-91) and early bound check (utils.c:546-567)
        const BLK_GLOBAL(double) *bu = blk_g(b.bu + (size_t)q * m), *bl = blk_g(b.bl + (size_t)q * m);
        const BLK_GLOBAL(int) *sin = b.sense_in ? blk_g(b.sense_in + (size_t)q * m) : nullptr;
        int bad = 0;
        for (int i = lane; i < m; i += 64) {
            int s = sin ? sin[i] : 0;
            if (s & DAQP_BINARY) bad |= 2;
            if (!(s & DAQP_IMMUTABLE)) {
                const double diff = bu[i] - bl[i];
                if (diff < -primal_tol) bad |= 1;
                else if (diff < zero_tol && !(s & DAQP_SOFT)) { s |= DAQP_ACTIVE + DAQP_IMMUTABLE; bad |= 4; }
            }
            sens[i] = s;
        }
        bad = (__any(bad & 2) ? 2 : 0) | (__any(bad & 1) ? 1 : 0) | (__any(bad & 4) ? 4 : 0);
        if (b.sense_in) activate = 1;
        if (bad & 4) activate = 1;
        if (bad & 2) flag = DAQP_EXIT_UNSUPPORTED;
        else if (bad & 1) flag = DAQP_EXIT_INFEASIBLE;
        if (force && flag > 0) flag = DAQP_NEEDS_SHIFT;   // forced proximal mode: the host starts with the shifted pass
        fl[lane] = (lane < n) ? blk_g(b.f)[(size_t)q * n + lane] : 0.0;
    }
    int diag = 0;
    blk_v4d X[NTT];
    if (flag > 0) {
        if (direct) copy_wait(); else stage_rows(Hs, H, n, n, n);
        WSYNC();
        blk_v4d T[NTT];
        int offd;
        blk_load<NT>(Hs, n, zero_tol, T, offd);
        SPROF(9);
        if (!__any(offd) && ms > 0) flag = DAQP_NEEDS_ORDERED;     // (a diagonal H with simple bounds: the RinvD branch scales them its own way, utils.c:284-312 -- the ordered kernel's)
        else if (!__any(offd)) {
            // RinvD_i = 1/sqrt(H_ii) (utils.c:245-312); a diagonal entry at or below zero_tol * max|H_ii| is shifted by the regularising
            // re-run and solved by the proximal outer loop
            const double hd = (lane < n) ? Hs[lane * n + lane] : 1.0;
            const double ha = hd < 0 ? -hd : hd;
            const double hscale = -wave_min((lane < n) ? -ha : 0.0);
            const double ftol = hscale > 0 ? zero_tol * hscale : zero_tol;
            const bool fail = lane < n && hd <= ftol;
            const int code = (b.st.eps_prox == 0.0 && hd <= zero_tol) ? DAQP_EXIT_NONCONVEX : DAQP_NEEDS_SHIFT;
            const unsigned long long fm = __ballot(fail);
            if (fm) flag = __builtin_amdgcn_readlane(code, __ffsll((long long)fm) - 1);   // the reference stops at the first such i
            else {
                WSYNC();
                xu[lane] = 1 / sqrt(hd);
                WSYNC();
                static_for<NT>([&](auto Ic) __attribute__((always_inline)) {
                    static_for<NT>([&](auto Jc) __attribute__((always_inline)) {
                        constexpr int I = Ic, J = Jc;
                        if constexpr (I <= J) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) X[blk_tix<NT>(I, J)][r] = (I == J && lk + 4 * r == lr) ? xu[16 * I + lr] : 0.0;
                        }
                    });
                });
                diag = 1;
            }
            WSYNC();
        } else {
            WSYNC();     // (every lane has taken its entries of H: the region is the factorisation's scratch now)
            double pmin = DAQP_INF, pmax = 0.0;
            const bool ok = blk_factor<NT>(T, n, zero_tol, X, Hs, pmin, pmax);
            // not clearly regular: the ordered kernel behind this launch decides (and shifts, or reports -5, as the reference would)
            if (!ok || !(pmin > 2.0 * zero_tol * pmax)) flag = DAQP_NEEDS_ORDERED;
            WSYNC();
        }
    }
    if (flag == DAQP_NEEDS_ORDERED) { if (lane == 0) qs->setup_flag = DAQP_NEEDS_ORDERED; return; }
    SPROF(0);
    int unc = 0;
    double vcol[NT];
    static_for<NT>([&](auto Jc) __attribute__((always_inline)) { vcol[Jc] = 0.0; });
    if (flag > 0) {
        // --- v = R^-T f (utils.c:474-497): column sums over this lane's rows, then over the four lane groups
        static_for<NT>([&](auto Jc) __attribute__((always_inline)) {
            constexpr int J = Jc;
            double p = 0;
            static_for<J + 1>([&](auto Ic) __attribute__((always_inline)) {
#pragma unroll
                for (int r = 0; r < 4; ++r) p = __builtin_fma(X[blk_tix<NT>(Ic, J)][r], fl[16 * Ic + lk + 4 * r], p);
            });
            p += __shfl_xor(p, 16);
            p += __shfl_xor(p, 32);
            vcol[J] = p;
            if (lk == 0 && 16 * J + lr < n) blk_g(b.v)[(size_t)q * n + 16 * J + lr] = p;
        });
        SPROF(1);
        // --- unconstrained optimum x = -R^-1 v (utils.c:618-662)
        if (mask & DAQP_UPDATE_unconstrained) {
            int fixed = 0;
            for (int i = lane; i < m; i += 64) fixed |= sens[i] & (DAQP_ACTIVE + DAQP_IMMUTABLE);
            if (!__any(fixed)) {
                unc = 1;
                static_for<NT>([&](auto Ic) __attribute__((always_inline)) {
                    constexpr int I = Ic;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        double p = 0;
                        static_for<NT - I>([&](auto Jj) __attribute__((always_inline)) { p = __builtin_fma(X[blk_tix<NT>(I, I + Jj)][r], vcol[I + Jj], p); });
                        p = blk_sum16(p);
                        if (lr == 0) xu[16 * I + lk + 4 * r] = -p;
                    }
                });
            }
        }
        WSYNC();
        if (mask & DAQP_UPDATE_eliminate) {   // eq_elim.c:127-164: that variant is not built
            int neq = 0;
            for (int base = 0; base < m; base += 64) {
                const int i = base + lane;
                const int isq = i < m && ((sens[i] & (DAQP_ACTIVE + DAQP_IMMUTABLE + DAQP_SOFT + DAQP_BINARY)) == (DAQP_ACTIVE + DAQP_IMMUTABLE));
                neq += __popcll(__ballot(isq));
            }
            if (neq > 5 && 10 * neq > n) flag = DAQP_EXIT_UNSUPPORTED;
        }
    }
    SPROF(2);
    int feasible = 1;
    if (flag > 0 && ms > 0) {
        // --- simple bounds (rows < ms of the LDP: the rows of R^-1, normalised: utils.c:569-585; their d: utils.c:499-544 / 664-676 + 151-159).
        // R^-1 goes through LDS once as a dense upper image (the tile region: H's image has been used up), lane <-> bound row from there.
        double *Rd = tile;
        for (int e = lane; e < n * n; e += 64) Rd[e] = 0.0;
        static_for<NT>([&](auto Jc) __attribute__((always_inline)) { if (lk == 0 && 16 * Jc + lr < n) vv[16 * Jc + lr] = vcol[Jc]; });
        WSYNC();
        static_for<NT>([&](auto Ic) __attribute__((always_inline)) {
            static_for<NT>([&](auto Jc) __attribute__((always_inline)) {
                constexpr int I = Ic, J = Jc;
                if constexpr (I <= J) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int i = 16 * I + lk + 4 * r, j = 16 * J + lr;
                        if (i < n && j < n && j >= i) Rd[i * n + j] = X[blk_tix<NT>(I, J)][r];
                    }
                }
            });
        });
        WSYNC();
        const BLK_GLOBAL(double) *bu = blk_g(b.bu + (size_t)q * m), *bl = blk_g(b.bl + (size_t)q * m);
        BLK_GLOBAL(double) *sc = blk_g(b.scaling + (size_t)q * m), *du = blk_g(b.dupper + (size_t)q * m), *dl = blk_g(b.dlower + (size_t)q * m);
        BLK_GLOBAL(blk_v2d) *Mq2 = blk_g(reinterpret_cast<blk_v2d *>(b.Mblk + (size_t)q * b.nblk * b.npair * 128));
        const bool own = lane < ms;
        const int i = own ? lane : 0;
        const double *a = Rd + i * n;
        double s0 = 0, s1 = 0, d0 = 0, d1 = 0;
        for (int j = 0; j + 1 < n; j += 2) {
            const double x0 = a[j], x1 = a[j + 1];
            s0 = __builtin_fma(x0, x0, s0); s1 = __builtin_fma(x1, x1, s1);
            d0 = __builtin_fma(x0, vv[j], d0); d1 = __builtin_fma(x1, vv[j + 1], d1);
        }
        if (n & 1) { const double x0 = a[n - 1]; s0 = __builtin_fma(x0, x0, s0); d0 = __builtin_fma(x0, vv[n - 1], d0); }
        const double scal = rsqrt(s0 + s1), draw = d0 + d1;
        WSYNC();
        fl[lane] = own ? scal : 1.0;           // (f has been used up: the rows' scalings for the packed image below)
        if (own) {
            const double bu_i = bu[i], bl_i = bl[i];
            sc[i] = scal;
            if (unc) {
                const double u0 = bu_i - xu[i], l0 = bl_i - xu[i];
                if (u0 < -primal_tol || l0 > primal_tol) feasible = 0;
                du[i] = u0 * scal; dl[i] = l0 * scal;
            } else {
                const double dsum = draw * scal;
                du[i] = bu_i * scal + dsum;
                dl[i] = bl_i * scal + dsum;
            }
            BLK_GLOBAL(blk_v2d) *dst = Mq2 + i;      // (row block 0: ms <= n <= 64)
            const int npair = b.npair;
            for (int t = 0; t < npair; ++t) {
                const double x0 = a[2 * t], x1 = (2 * t + 1 < n) ? a[2 * t + 1] : 0.0;
                dst[(size_t)t * 64] = (blk_v2d){x0 * scal, x1 * scal};
            }
        }
        WSYNC();
    }
    if (flag > 0) {
        // packed upper image of R^-1 for the solve kernel / warm updates, straight from the tiles (rows < ms normalised: the reference keeps them so)
        BLK_GLOBAL(double) *Rp = blk_g(b.Rinv + (size_t)q * b.rtri);
        static_for<NT>([&](auto Ic) __attribute__((always_inline)) {
            static_for<NT>([&](auto Jc) __attribute__((always_inline)) {
                constexpr int I = Ic, J = Jc;
                if constexpr (I <= J) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int i = 16 * I + lk + 4 * r, j = 16 * J + lr;
                        double x = X[blk_tix<NT>(I, J)][r];
                        if (ms > 0 && i < ms) x *= fl[i];
                        if (i < n && j < n && j >= i) Rp[roff(i, n) + j] = x;
                    }
                }
            });
        });
        WSYNC();
    }
    SPROF(6);
    // --- general rows, TR at a time through the LDS tile (which takes the place of H's image)
    if (flag > 0) {
        const BLK_GLOBAL(double) *bu = blk_g(b.bu + (size_t)q * m), *bl = blk_g(b.bl + (size_t)q * m);
        BLK_GLOBAL(double) *sc = blk_g(b.scaling + (size_t)q * m), *du = blk_g(b.dupper + (size_t)q * m), *dl = blk_g(b.dlower + (size_t)q * m);   // HBM
        BLK_GLOBAL(blk_v2d) *Mq2 = blk_g(reinterpret_cast<blk_v2d *>(b.Mblk + (size_t)q * b.nblk * b.npair * 128));
        const int npair = b.npair;
        WSYNC();
        if (direct && mA > 0) copy_async(tile, A, (mA < TR ? mA : TR) * n);
        for (int tb = 0; tb < mA && flag > 0; tb += TR) {
            const int rows = (mA - tb) < TR ? (mA - tb) : TR;
            if (direct) copy_wait();
            else { WSYNC(); stage_rows(tile, A + (size_t)tb * n, rows, n, ldr); }
            WSYNC();
            SPROF(7);
            const bool own = lane < rows;
            const double *a = tile + (own ? lane : 0) * ldr;
            const int gi = ms + (own ? tb + lane : tb);
            const double bu_gi = bu[gi], bl_gi = bl[gi];   // issued now, used after the tile's arithmetic: no exposed trip to HBM
            double sunc = 0;
            if (unc) sunc = chain_add8(0.0, n, [&](int j) { return a[j]; }, [&](int j) { return xu[j]; });
            // M = A R^-1 on the matrix cores: one 16-row tile of A against the resident tiles of R^-1 (A: lane l supplies
            // A[l&15][4kt + (l>>4)]; B: register kt & 3 of tile (kt >> 2, ct); D: column l&15, row (l>>4) + 4 reg).  Still in that
            // layout: the rows' sums of squares and their products with v -- four registers and a sixteen-lane sum each, where the
            // lane <-> row pass ran two dependent chains of n multiply-adds per row.
            for (int rt = 0; rt * 16 < rows; ++rt) {
                blk_v4d acc4[CT];
                static_for<CT>([&](auto ct) __attribute__((always_inline)) { acc4[ct] = (blk_v4d){0.0, 0.0, 0.0, 0.0}; });
                const int arow = rt * 16 + lr;
                const bool rowok = arow < rows;
                const double *arowp = tile + (rowok ? arow : 0) * ldr;
                double av[KT];
                static_for<KT>([&](auto kt) __attribute__((always_inline)) {
                    const int kk = 4 * kt + lk;
                    const bool kok = kk < n;
                    const double aload = arowp[kok ? kk : 0];
                    av[kt] = (rowok && kok) ? aload : 0.0;
                });
                static_for<KT>([&](auto kt) __attribute__((always_inline)) {
                    static_for<CT>([&](auto ct) __attribute__((always_inline)) {
                        if constexpr ((kt >> 2) <= ct)
                            acc4[ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[kt], X[blk_tix<NT>(kt >> 2, ct)][kt & 3], acc4[ct], 0, 0, 0);
                    });
                });
                double tailv[4] = {0.0, 0.0, 0.0, 0.0};
                if constexpr (TAIL) {
                    static_for<4>([&](auto tc) __attribute__((always_inline)) {
                        constexpr int t = tc;
                        if (t < tailc) {        // (wave-uniform)
                            double p = 0;
                            static_for<KT>([&](auto kt) __attribute__((always_inline)) {
                                p = __builtin_fma(av[kt], blk_bcast<t>(X[blk_tix<NT>(kt >> 2, NT - 1)][kt & 3]), p);
                            });
                            p += __shfl_xor(p, 16);
                            p += __shfl_xor(p, 32);
                            tailv[t] = p;     // M[row lr of this row tile][16 (NT - 1) + t], in every lane group
                        }
                    });
                }
                double s4[4], d4[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    double s = 0, dd = 0;
                    static_for<CT>([&](auto ct) __attribute__((always_inline)) {
                        s = __builtin_fma(acc4[ct][r], acc4[ct][r], s);
                        dd = __builtin_fma(acc4[ct][r], vcol[ct], dd);
                    });
                    s4[r] = blk_sum16(s); d4[r] = blk_sum16(dd);
                }
                WSYNC();   // every lane's reads of this row tile precede the in-place overwrite below
                static_for<CT>([&](auto ct) __attribute__((always_inline)) {
                    static_for<4>([&](auto r) __attribute__((always_inline)) {
                        const int row = rt * 16 + lk + 4 * r, col = 16 * ct + lr;
                        if (col < n && row < rows) tile[row * ldr + col] = acc4[ct][(int)r];
                    });
                });
                if (lr == 0) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) { srow[rt * 16 + lk + 4 * r] = s4[r]; drow[rt * 16 + lk + 4 * r] = d4[r]; }
                }
                if constexpr (TAIL) {
                    static_for<4>([&](auto tc) __attribute__((always_inline)) {
                        if (tc < tailc && lk == 0 && rowok) tile[arow * ldr + 16 * (NT - 1) + tc] = tailv[tc];
                    });
                }
            }
            WSYNC();
            SPROF(8);
            // lane <-> row: normalise (utils.c:586-613), d (utils.c:499-544 / 664-676 + 151-159), blocked store
            double s = srow[own ? lane : 0], draw = drow[own ? lane : 0];
            if constexpr (TAIL) {
                static_for<4>([&](auto tc) __attribute__((always_inline)) {
                    if (tc < tailc) {
                        const double e = a[16 * (NT - 1) + tc];
                        s = __builtin_fma(e, e, s);
                        draw = __builtin_fma(e, blk_bcast<tc>(vcol[NT - 1]), draw);
                    }
                });
            }
            double scal = 1.0;
            int rowbad = 0;
            if (own) {
                if (s < zero_tol) {
                    if (bu_gi < -zero_tol || bl_gi > zero_tol)
                        if (!(sens[gi] & DAQP_IMMUTABLE) && !(sens[gi] & DAQP_SOFT)) rowbad = 1;
                    sens[gi] = DAQP_IMMUTABLE;
                } else scal = rsqrt(s);
                sc[gi] = scal;
                if (unc) {
                    const double u0 = bu_gi - sunc, l0 = bl_gi - sunc;
                    if (u0 < -primal_tol || l0 > primal_tol) feasible = 0;
                    du[gi] = u0 * scal; dl[gi] = l0 * scal;
                } else {
                    const double dsum = draw * scal;
                    du[gi] = bu_gi * scal + dsum;
                    dl[gi] = bl_gi * scal + dsum;
                }
            }
            SPROF(3);
            {
                BLK_GLOBAL(blk_v2d) *dst = Mq2 + ((size_t)(gi >> 6) * npair) * 64 + (gi & 63);
                // eight column pairs in flight; the pairs past the row's end repeat its last one (the same bytes stored again: no
                // branch between the loads -- with one, every load waited for the one before it: 25 LDS round trips per row and tile)
                if (own) {
                    if (direct) {
                        const double2 *a2 = reinterpret_cast<const double2 *>(a);
                        for (int tq = 0; tq < npair; tq += 8) {
                            double2 v8[8];
#pragma unroll
                            for (int u = 0; u < 8; ++u) v8[u] = a2[(tq + u < npair) ? tq + u : npair - 1];
#pragma unroll
                            for (int u = 0; u < 8; ++u) dst[(size_t)((tq + u < npair) ? tq + u : npair - 1) * 64] = (blk_v2d){v8[u].x * scal, v8[u].y * scal};
                        }
                    } else {
                        for (int tq = 0; tq < npair; tq += 8) {
                            double vx[8], vy[8];
#pragma unroll
                            for (int u = 0; u < 8; ++u) {
                                const int t = (tq + u < npair) ? tq + u : npair - 1;
                                vx[u] = a[2 * t];
                                const double y = a[(2 * t + 1 < n) ? 2 * t + 1 : 2 * t];
                                vy[u] = (2 * t + 1 < n) ? y : 0.0;
                            }
#pragma unroll
                            for (int u = 0; u < 8; ++u) dst[(size_t)((tq + u < npair) ? tq + u : npair - 1) * 64] = (blk_v2d){vx[u] * scal, vy[u] * scal};
                        }
                    }
                }
            }
            if (direct && tb + TR < mA) {   // the tile has been read: the next one loads while this one's stores drain
                WSYNC();
                copy_async(tile, A + (size_t)(tb + TR) * n, ((mA - tb - TR) < TR ? (mA - tb - TR) : TR) * n);
            }
            if (__any(rowbad)) flag = DAQP_EXIT_INFEASIBLE;
            SPROF(4);
        }
    }
    const int all_feasible = __all(feasible);
    int sing = kEmpty;
    if (flag > 0 && unc && all_feasible) { sing = DAQP_UNCONSTRAINED_OPTIMAL; activate = 0; }
    if (flag > 0) {
        if (lane < n && unc) blk_g(b.xunc)[(size_t)q * n + lane] = xu[lane];
    }
    for (int i = lane; i < m; i += 64) blk_g(b.sense)[(size_t)q * m + i] = sens[i];
    SPROF(5);
    if (lane == 0) {
        qs->n_active = 0; qs->reuse_ind = 0; qs->sing_ind = sing; qs->iterations = 0;
        qs->lam_swapped = 0; qs->setup_flag = flag; qs->need_activate = (flag > 0) ? activate : 0; qs->pad_ = 0;
        qs->exitflag = flag; qs->fval = 0; qs->soft_slack = 0; qs->diag_h = diag; qs->n_prox = 0;
        qs->upd_flag = 0;
        if (kProfile && b.prof) for (int i = 0; i < 10; ++i) blk_g(b.prof)[(size_t)q * 32 + i] = pt[i];
    }
#undef SPROF
}

// the instantiations: (blocks, column width, tail columns on the vector pipe) -- the first that fits n is taken
#define DAQP_BLK_SHAPES \
    DAQP_BLK_SHAPE(2, 24, true) DAQP_BLK_SHAPE(2, 32, false) DAQP_BLK_SHAPE(3, 40, true) DAQP_BLK_SHAPE(3, 48, false) \
    DAQP_BLK_SHAPE(4, 56, true) DAQP_BLK_SHAPE(4, 56, false) DAQP_BLK_SHAPE(4, 64, false)

} // namespace daqp_amd
