// setup_fast.hip.h -- QP -> LDP for n <= NMAX <= 64, one wavefront per QP.  Arithmetic (operation
// ORDER included) is exactly that of k_setup / reference src/utils.c:223-687; only the schedule differs:
//   * Cholesky (utils.c:335-352) and R -> R^-1 (utils.c:380-389) are FUSED and RIGHT-looking, entirely in
//     registers: lane j holds column j of the trailing Schur complement (c[]) and lane k holds row k of the
//     running inverse accumulators (a[]).  Step k finishes row k of R, broadcasts its entries with
//     v_readlane and applies   c_ij -= r_ki r_kj   and   a_kj -= r_ij t_i   to every later column.  Each
//     element still receives its subtractions in ascending k, i.e. the reference's left-looking chains
//     bit for bit, but no step waits for LDS and the two updates share the broadcasts.  Writing the result
//     of step k one register lower (c[i] = c[i+1] - ...) keeps "row k" in register 0, so the loop over k
//     stays rolled with compile-time register indices.
//   * M = A R^-1 (utils.c:434-472): default on the matrix cores (v_mfma_f64_16x16x4_f64) with the R^-1
//     fragments register-resident for the whole problem; with exact_setup lane <-> row of A, r-OUTER with
//     all n accumulators of the row in registers (terms arrive diagonal first, then decreasing r).
#pragma once
#include "wave_ldp_reg.hip.h"
#include "batch_dev.hip.h"

namespace daqp_amd {

// acc -= sum_{k<cnt} A(k)*B(k), k ascending; loads of each group of 8 issued before its chain
template <class FA, class FB>
__device__ __forceinline__ double chain_sub8(double acc, int cnt, FA A, FB B)
{
    int k0 = 0;
    for (; k0 + 8 <= cnt; k0 += 8) {
        double a[8], c[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) { a[q] = A(k0 + q); c[q] = B(k0 + q); }
#pragma unroll
        for (int q = 0; q < 8; ++q) acc -= a[q] * c[q];
    }
    if (k0 < cnt) {
        double a[8], c[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const bool in = k0 + q < cnt;
            a[q] = in ? A(in ? k0 + q : k0) : 0.0;
            c[q] = in ? B(in ? k0 + q : k0) : 0.0;
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) if (k0 + q < cnt) acc -= a[q] * c[q];
    }
    return acc;
}
// acc += sum A(k)*B(k), k ascending
template <class FA, class FB>
__device__ __forceinline__ double chain_add8(double acc, int cnt, FA A, FB B)
{
    int k0 = 0;
    for (; k0 + 8 <= cnt; k0 += 8) {
        double a[8], c[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) { a[q] = A(k0 + q); c[q] = B(k0 + q); }
#pragma unroll
        for (int q = 0; q < 8; ++q) acc += a[q] * c[q];
    }
    for (; k0 < cnt; ++k0) acc += A(k0) * B(k0);
    return acc;
}

// LDS: [ Rsq: n x nsq square R^-1 (zeros left of the diagonal; first the staged H) | tile (64 x ldr: A rows,
// overwritten by their M rows) | f v xu | scaling dupper dlower | sense ].  Without exact_setup the R^-1
// fragments live in registers during the M phase, so the tile aliases Rsq and four workgroups fit a CU.
struct FastLds { int R, fv, vv, xu, sc, du, dl, tile, sens, nsq, total_bytes; };
__host__ __device__ inline int fast_nsq(int n) { int q = n < 2 ? 2 : n; while ((q & 3) != 2) ++q; return q; }
// rows of A per LDS tile: 56 when that does not add a tile (150 general rows: 56 + 56 + 38 like 64 + 64 + 22).  With d and the
// scalings of the general rows written straight to HBM it brings a C2 workgroup from 31 KB of LDS to 24 KB: 6 per CU instead of
// 5 -- the kernel's phases are latency-bound per wave, resident waves are what counts.  (Measured: balanced tiles of 52 rows,
// 7 workgroups per CU, are slower again -- 8.3 against 8.0 ms per 100 k QPs.)
__host__ __device__ inline int fast_tile_rows(int mA) { return (mA > 0 && (mA + 55) / 56 == (mA + 63) / 64) ? 56 : 64; }
__host__ __device__ inline FastLds fast_lds(int n, int m, int exact, int mA = -1)
{
    FastLds s;
    const int nsq = fast_nsq(n), np = round_up(n, 2), mp = round_up(m, 2), ldr = n | 1;   // tile stride: n | 1 (staged) or n (direct copy)
    const int TR = mA < 0 ? 64 : fast_tile_rows(mA);
    const int rsz = round_up(n * nsq + 8, 2), tsz = round_up(TR * ldr, 2);
    (void)mp;
    int o = 0;
    s.nsq = nsq;
    s.R = 0;
    if (exact) { s.tile = rsz; o = rsz + tsz; } else { s.tile = 0; o = rsz > tsz ? rsz : tsz; }
    s.fv = o; o += np; s.vv = o; o += np; s.xu = o; o += np;
    s.sc = o; o += (mA < 0) ? np : round_up(m - mA < np ? m - mA : np, 2); s.du = s.dl = o;   // (scaling of the simple bounds only: ms <= n entries)
    s.sens = o;
    s.total_bytes = o * 8 + round_up(m, 4) * 4;
    return s;
}

// PROX = true: the regularising re-run of the problems flagged DAQP_NEEDS_SHIFT (utils.c:354-377, see k_setup): the same
// code with H + hshift[q] on the diagonal (a diagonal H: in its singular coordinates only), the stricter pivot ratio and
// the list of shifted coordinates written out.  A separate instantiation, so that the ordinary pass keeps its code.
// FM = true (default arithmetic; never together with PROX): the updates of the fused Cholesky / inverse sweep as fused
// multiply-adds -- five instead of seven instructions per updated pair, results equal to ~1e-16 relative like the rest of the
// default mode's setup (M comes off the matrix cores there anyway).
template <int NMAX, bool PROX = false, bool FM = false>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(NMAX >= 56 ? 2 : 1, NMAX >= 56 ? 2 : 8))) void k_setup_fast(BatchDev b, int mask)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int q = blockIdx.x, lane = lane_id();
    const int n = b.n, m = b.m, ms = b.ms, mA = b.mA;
    const FastLds o = fast_lds(n, m, b.exact_setup, mA);
    const int TR = fast_tile_rows(mA);
    const int nsq = o.nsq;
    double *Rsq = smem + o.R, *fl = smem + o.fv, *vv = smem + o.vv, *xu = smem + o.xu;
    double *scs = smem + o.sc, *tile = smem + o.tile;
    double *sc = b.scaling + (size_t)q * m, *du = b.dupper + (size_t)q * m, *dl = b.dlower + (size_t)q * m;   // HBM
    int *sens = reinterpret_cast<int *>(smem + o.sens);
    const double *H = b.H + (size_t)q * n * n, *f = b.f + (size_t)q * n, *A = b.A + (size_t)q * mA * n;
    const double *bu = b.bu + (size_t)q * m, *bl = b.bl + (size_t)q * m;
    const DAQPSettings &st = b.st;
    QState *qs = b.qs + q;
    double shift = 0.0;
    int nprox = 0;
    const bool force = st.eps_prox > 0.0;
    if constexpr (PROX) {
        if (__builtin_amdgcn_readfirstlane(qs->setup_flag) != DAQP_NEEDS_SHIFT) return;
        shift = b.hshift[q];
    } else if (mask & kSetupOnlyMarked) {      // behind k_setup_blk: the problems it left to the ordered factorisation, nobody else
        if (__builtin_amdgcn_readfirstlane(qs->setup_flag) != DAQP_NEEDS_ORDERED) return;
    }
    // rows of even length (and not a multiple of 32 doubles, which would put every row of the tile on the same
    // LDS banks) are copied HBM -> LDS directly, unpadded; everything else is staged through registers with an odd stride
    const bool direct = !(n & 1) && (n & 31) && !(((size_t)H | (size_t)A) & 15);
    const int ldr = direct ? n : (n | 1);
    int flag = 1, activate = 0;
    long long pt[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    long long t0 = (kProfile && b.prof) ? (long long)__builtin_readcyclecounter() : 0;
#define SPROF(slot) do { if (kProfile && b.prof) { const long long t1 = (long long)__builtin_readcyclecounter(); pt[slot] += t1 - t0; t0 = t1; } } while (0)

    if (direct) copy_async(Rsq, H, n * n);     // in flight while the bounds are checked
    // --- sense (utils.c:84-91) and early bound check (utils.c:546-567)
    int bad = 0;
    for (int i = lane; i < m; i += 64) {
        int s = b.sense_in ? b.sense_in[(size_t)q * m + i] : 0;
        if (s & DAQP_BINARY) bad |= 2;
        if (!(s & DAQP_IMMUTABLE)) {
            const double diff = bu[i] - bl[i];
            if (diff < -st.primal_tol) bad |= 1;
            else if (diff < st.zero_tol && !(s & DAQP_SOFT)) { s |= DAQP_ACTIVE + DAQP_IMMUTABLE; bad |= 4; }
        }
        sens[i] = s;
    }
    bad = (__any(bad & 2) ? 2 : 0) | (__any(bad & 1) ? 1 : 0) | (__any(bad & 4) ? 4 : 0);
    if (b.sense_in) activate = 1;
    if (bad & 4) activate = 1;
    if (bad & 2) flag = DAQP_EXIT_UNSUPPORTED;
    else if (bad & 1) flag = DAQP_EXIT_INFEASIBLE;
    if (!PROX && force && flag > 0) flag = DAQP_NEEDS_SHIFT;   // forced proximal mode: the host starts with the shifted pass
    if (lane < n) fl[lane] = f[lane];
    // --- 1/2 (H + H') (utils.c:318-324) into registers: lane j <-> column j, c[i] = row i
    double pmin = DAQP_INF, pmax = 0.0;
    int diag = 0;        // H diagonal: the reference's RinvD branch (utils.c:245-312)
    double hsq = 0;      // sqrt(H_ii) of this lane's coordinate in that branch
    if (flag > 0) {
        double c[NMAX], a[NMAX];
        if (direct) copy_wait(); else stage_rows(Rsq, H, n, n, n);
        WSYNC();
        int offd = 0;    // an entry above the diagonal of this lane's column exceeds zero_tol (utils.c:245-252 looks at those only)
        {
            const int jj = lane < n ? lane : 0;
            static_for<NMAX / 8>([&](auto g) __attribute__((always_inline)) {
                if (8 * g < n) {
                    static_for<8>([&](auto h) __attribute__((always_inline)) {
                        constexpr int i = 8 * g + h;
                        const int ii = i < n ? i : 0;
                        const double hij = Rsq[ii * n + jj], hji = Rsq[jj * n + ii];
                        const double val = (jj == ii) ? (PROX ? hij + shift : hij) : 0.5 * (hij + hji);
                        if constexpr (i == 0) offd = (lane > 0 && lane < n && (hij > st.zero_tol || hij < -st.zero_tol)) ? 1 : 0;
                        c[i] = (lane < n && i < n) ? val : 0.0;
                        a[i] = 0.0;
                    });
                } else {
                    static_for<8>([&](auto h) __attribute__((always_inline)) { c[8 * g + h] = 0.0; a[8 * g + h] = 0.0; });
                }
            });
        }
        // offd so far: row 0 only (a dense H is dismissed here at the price of one compare); if row 0 passes, look at
        // every entry above the diagonal of this lane's column
        bool isdiag = !__any(offd);
        if (isdiag) {
            for (int i = 1; i < n; ++i) {
                const double hij = Rsq[i * n + (lane < n ? lane : 0)];
                offd |= (lane < n && i < lane && (hij > st.zero_tol || hij < -st.zero_tol)) ? 1 : 0;
            }
            isdiag = !__any(offd);
        }
        if (isdiag) {
            // RinvD_i = 1/sqrt(H_ii), scaling_i = sqrt(H_ii) for simple bounds; a diagonal entry at or below
            // zero_tol * max|H_ii| is shifted by the regularising re-run of k_setup and solved by the proximal outer loop
            double hd = (lane < n) ? Rsq[lane * n + lane] : 1.0;
            const double ha = hd < 0 ? -hd : hd;
            const double hscale = -wave_min((lane < n) ? -ha : 0.0);
            const double ftol = hscale > 0 ? st.zero_tol * hscale : st.zero_tol;
            bool fail = lane < n && hd <= ftol;
            int code = (st.eps_prox == 0.0 && hd <= st.zero_tol) ? DAQP_EXIT_NONCONVEX : DAQP_NEEDS_SHIFT;
            if constexpr (PROX) {   // semi-proximal: the singular coordinates only, remembered in prox_mask (utils.c:294-303)
                const bool low = lane < n && (fail || force);
                if (low) hd += shift;
                if (lane < n) b.prox_mask[(size_t)q * n + lane] = low ? 1 : 0;
                nprox = __popcll(__ballot(low));
                fail = lane < n && hd <= st.zero_tol;
                code = DAQP_EXIT_NONCONVEX;
            }
            const unsigned long long fm = __ballot(fail);
            if (fm) flag = __builtin_amdgcn_readlane(code, __ffsll((long long)fm) - 1);   // the reference stops at the first such i
            else {
                hsq = sqrt(hd);
                const double rinvd = 1 / hsq;
                WSYNC();
                for (int e = lane; e < n * nsq; e += 64) Rsq[e] = 0.0;
                WSYNC();
                if (lane < n) Rsq[lane * nsq + lane] = rinvd;
                diag = 1;
            }
        }
        WSYNC();
        if (lane < 8) Rsq[n * nsq + lane] = 0.0;   // padding behind the last row (read, never used, by the 8-column groups)
        // --- fused Cholesky (utils.c:335-352) and R -> R^-1 (utils.c:380-389).  Entering step k, c[i] is row
        // k+i of the Schur complement (lane <-> column) and, in lanes < k, a[i] is the accumulator of
        // (R^-1)[lane][k+i]; lanes >= k hold +-0 there.
        // Phase P covers the steps k that still have 8(P-1) < n-1-k <= 8P later rows: its body updates exactly
        // 8P registers with no branch inside (rows >= n only ever hold zeros or unused values).
        static_for<NMAX / 8 + 1>([&](auto pp) __attribute__((always_inline)) {
            constexpr int P = NMAX / 8 - pp;
            int k = n - 1 - 8 * P;
            if (k < 0) k = 0;
            const int kend = isdiag ? 0 : ((P == 0) ? n : n - 1 - 8 * (P - 1));
            for (; k < kend && flag > 0; ++k) {
                const double dg = rl(c[0], k);
                if (dg <= st.zero_tol) { flag = (!PROX && st.eps_prox == 0.0) ? DAQP_EXIT_NONCONVEX : DAQP_NEEDS_SHIFT; break; }
                if (dg < pmin) pmin = dg;
                if (dg > pmax) pmax = dg;
                const double dgi = 1 / sqrt(dg);
                const double rk = (lane > k) ? c[0] * dgi : 0.0;                   // r_kj, j > k
                const double tk = a[0] * dgi;                                      // t_k *= 1/r_kk for the rows above
                const double col = (lane < k) ? tk : ((lane == k) ? dgi : 0.0);    // column k of R^-1 (zeros below the diagonal)
                if (lane < n) Rsq[lane * nsq + k] = col;
                // lane k starts its own row now: -0.0 - r_kj*(1/r_kk) == r_kj * -(1/r_kk) bit for bit, so the start is
                // the common update applied to a forced -0.0 (lanes >= k only ever hold +-0: low word 0, OR the sign in)
                const unsigned sgn = (lane >= k) ? 0x80000000u : 0u;
                // rotate row k so that r_{k,k+1+i} sits in lane i: a v_readlane with an immediate lane costs half of
                // one with a computed lane (s_add + SGPR-index hazard), and there are ~n^2/2 of them
                const double rot = __shfl(rk, (lane + k + 1) & 63);
                static_for<8 * P>([&](auto ii) __attribute__((always_inline)) {
                    constexpr int i = ii;
                    if constexpr (i + 1 < NMAX) {
                        const double s = rl(rot, i);                               // r_{k,k+1+i}, wave-uniform
                        const double ap = __hiloint2double(__double2hiint(a[i + 1]) | sgn, __double2loint(a[i + 1]));
                        if constexpr (FM) { c[i] = __builtin_fma(-s, rk, c[i + 1]); a[i] = __builtin_fma(-s, col, ap); }
                        else { c[i] = c[i + 1] - s * rk; a[i] = ap - s * col; }
                    }
                });
            }
        });
        if (flag > 0 && !isdiag && pmin <= ((PROX && !force) ? sqrt(st.zero_tol) : st.zero_tol) * pmax)   // utils.c:354-356
            flag = (!PROX && st.eps_prox == 0.0) ? DAQP_EXIT_NONCONVEX : DAQP_NEEDS_SHIFT;
        if constexpr (PROX) {
            if (flag > 0 && !isdiag) {
                nprox = n;
                if (lane < n) b.prox_mask[(size_t)q * n + lane] = 1;
            }
        }
        WSYNC();
    }
    SPROF(0);
    if (flag > 0) {
        // --- v = R^-T f (utils.c:474-497)
        if (lane < n) {
            const int i = lane;
            double acc = Rsq[i * nsq + i] * fl[i];
            acc = chain_add8(acc, i, [&](int t) { return Rsq[(i - 1 - t) * nsq + i]; }, [&](int t) { return fl[i - 1 - t]; });
            vv[i] = acc;
        }
        WSYNC();
    }
    SPROF(1);
    // --- unconstrained optimum (utils.c:618-662)
    int unc = 0;
    if (flag > 0 && (mask & DAQP_UPDATE_unconstrained)) {
        int fixed = 0;
        for (int i = lane; i < m; i += 64) fixed |= sens[i] & (DAQP_ACTIVE + DAQP_IMMUTABLE);
        if (!__any(fixed)) {
            unc = 1;
            if (lane < n) {
                const int i = lane;
                const double s = chain_add8(0.0, n - i, [&](int t) { return Rsq[i * nsq + i + t]; }, [&](int t) { return vv[i + t]; });
                xu[i] = -s;
            }
            WSYNC();
        }
    }
    if (flag > 0 && (mask & DAQP_UPDATE_eliminate)) {   // eq_elim.c:127-164: that variant is not built
        int neq = 0;
        for (int base = ms; base < m; base += 64) {
            const int i = base + lane;
            const int isq = i < m && ((sens[i] & (DAQP_ACTIVE + DAQP_IMMUTABLE + DAQP_SOFT + DAQP_BINARY)) == (DAQP_ACTIVE + DAQP_IMMUTABLE));
            neq += __popcll(__ballot(isq));
        }
        if (neq > 5 && 10 * neq > n) flag = DAQP_EXIT_UNSUPPORTED;
    }
    SPROF(2);

    // --- simple bounds: rows < ms of R^-1 normalised (utils.c:569-585), their d, their dense image in M.
    // Rsq itself stays unscaled (the general rows below multiply by the unscaled inverse, as the
    // reference does at that point); every consumer applies the one multiplication by s itself.
    int feasible = 1;
    double *Mq = b.Mblk + (size_t)q * b.nblk * b.npair * 128;
    if (flag > 0) {
        if (lane < ms) {
            const int i = lane;
            const double *Ri = Rsq + i * nsq;
            double s = 0;
            if (diag) s = hsq;   // scaling_i = sqrt(H_ii) (utils.c:309); the row of R^-1 counts as the unit vector
            else {
                for (int j = i; j < n; ++j) s += Ri[j] * Ri[j];
                s = 1 / sqrt(s);
            }
            sc[i] = s; scs[i] = s;
            if (unc) {
                const double u0 = bu[i] - xu[i], l0 = bl[i] - xu[i];
                if (u0 < -st.primal_tol || l0 > st.primal_tol) feasible = 0;
                du[i] = u0 * s; dl[i] = l0 * s;
            } else {
                double t = 0;
                if (diag) t = vv[i];   // utils.c:527-531
                else for (int j = i; j < n; ++j) t += (Ri[j] * s) * vv[j];
                du[i] = bu[i] * s + t;
                dl[i] = bl[i] * s + t;
            }
            double2 *dst = reinterpret_cast<double2 *>(Mq) + ((size_t)(i >> 6) * b.npair) * 64 + (i & 63);
            for (int t = 0; t < b.npair; ++t) {
                double2 vpair;
                if (diag) { vpair.x = (2 * t == i) ? 1.0 : 0.0; vpair.y = (2 * t + 1 == i) ? 1.0 : 0.0; }
                else {
                    vpair.x = (2 * t >= i) ? Ri[2 * t] * s : 0.0;
                    vpair.y = (2 * t + 1 >= i && 2 * t + 1 < n) ? Ri[2 * t + 1] * s : 0.0;
                }
                dst[(size_t)t * 64] = vpair;
            }
        }
        WSYNC();
        // packed upper image for the solve kernel / warm updates (rows < ms normalised)
        double *Rp = b.Rinv + (size_t)q * b.rtri;
        for (int i = 0; i < n; ++i) {
            const int j = i + lane;
            if (j < n) {
                double val = Rsq[i * nsq + j];
                if (i < ms && !diag) val *= scs[i];
                Rp[roff(i, n) + j] = val;
            }
        }
    }
    // R^-1 fragments of the MFMA path: Bf[kt][ct] = R^-1[4kt + (lane>>4)][16ct + (lane&15)], upper triangular
    // (k tiles only up to the column tile's last column), zero outside n x n
    constexpr int KT = NMAX / 4, CT = (NMAX + 15) / 16;
    double Bf[KT][CT];
    if (flag > 0 && !b.exact_setup) {
        const int lr = lane & 15, lk = lane >> 4;
        static_for<KT>([&](auto kt) __attribute__((always_inline)) {
            static_for<CT>([&](auto ct) __attribute__((always_inline)) {
                if constexpr (kt <= 4 * ct + 3) {
                    const int kk = 4 * kt + lk, cc = 16 * ct + lr;
                    const bool ok = kk < n && cc < n;
                    const double bload = Rsq[(kk < n ? kk : 0) * nsq + (cc < n ? cc : 0)];
                    Bf[kt][ct] = ok ? bload : 0.0;
                } else Bf[kt][ct] = 0.0;
            });
        });
    }
    SPROF(6);
    // --- general rows, 64 at a time through the LDS tile; lane <-> row
    if (flag > 0) {
        WSYNC();   // Rsq (under the tile unless exact_setup) has no readers left
        if (direct && mA > 0) copy_async(tile, A, (mA < TR ? mA : TR) * n);
        for (int tb = 0; tb < mA && flag > 0; tb += TR) {
            const int rows = (mA - tb) < TR ? (mA - tb) : TR;
            if (direct) copy_wait();
            else { WSYNC(); stage_rows(tile, A + (size_t)tb * n, rows, n, ldr); }
            WSYNC();
            SPROF(7);
            const int k = tb + lane;
            const bool own = lane < rows;
            const double *a = tile + (own ? lane : 0) * ldr;
            const int gi = ms + (own ? k : tb);
            const double bu_gi = bu[gi], bl_gi = bl[gi];   // issued now, used after the tile's arithmetic: no exposed trip to HBM
            double sunc = 0;
            if (unc) sunc = chain_add8(0.0, n, [&](int j) { return a[j]; }, [&](int j) { return xu[j]; });
            double acc[NMAX];
            if (b.exact_setup) {
                // M row: r-outer, accumulators in registers (utils.c:441-453: diagonal term first, then decreasing r)
                static_for<NMAX>([&](auto c) __attribute__((always_inline)) { acc[c] = 0.0; });
                for (int r = n - 1; r >= 0; --r) {
                    const double ar = a[r];
                    const double2 *Rr = reinterpret_cast<const double2 *>(Rsq + r * nsq);
                    // groups of 8 columns, branch-free inside: entries left of the diagonal are stored zeros,
                    // so they add +0.0; only groups that reach the diagonal or beyond are executed
                    static_for<NMAX / 8>([&](auto g) __attribute__((always_inline)) {
                        if (8 * g + 7 >= r && 8 * g < n) {
                            static_for<4>([&](auto h) __attribute__((always_inline)) {
                                const double2 rv = Rr[4 * g + h];
                                acc[8 * g + 2 * h] += rv.x * ar;
                                acc[8 * g + 2 * h + 1] += rv.y * ar;
                            });
                        }
                    });
                }
            } else {
                // M = A R^-1 on the matrix cores: v_mfma_f64_16x16x4_f64, one 16-row tile of A against the
                // register-resident fragments of R^-1.  A: lane l supplies A[l&15][4kt + (l>>4)], B: R^-1[4kt + (l>>4)][16ct + (l&15)],
                // D: col = l&15, row = (l>>4) + 4*reg.  No run-time guards inside: fragments beyond n are zeros.
                // The summation order differs from the reference's (fp64, fused): M agrees to ~1e-16 relative.
                typedef double v4d __attribute__((ext_vector_type(4)));
                const int lr = lane & 15, lk = lane >> 4;
                for (int rt = 0; rt * 16 < rows; ++rt) {
                    v4d acc4[CT];
                    static_for<CT>([&](auto ct) __attribute__((always_inline)) { acc4[ct] = (v4d){0.0, 0.0, 0.0, 0.0}; });
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
                            if constexpr (kt <= 4 * ct + 3)
                                acc4[ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[kt], Bf[kt][ct], acc4[ct], 0, 0, 0);
                        });
                    });
                    WSYNC();   // every lane's reads of this row tile precede the in-place overwrite below
                    static_for<CT>([&](auto ct) __attribute__((always_inline)) {
                        static_for<4>([&](auto r) __attribute__((always_inline)) {
                            const int row = rt * 16 + lk + 4 * r, col = 16 * ct + lr;
                            if (col < n && row < rows) tile[row * ldr + col] = acc4[ct][(int)r];
                        });
                    });
                }
                WSYNC();
                SPROF(8);
                if (direct) {   // even row length, 16-byte aligned rows: two columns per LDS read
                    const double2 *a2 = reinterpret_cast<const double2 *>(a);
                    static_for<NMAX / 2>([&](auto t) __attribute__((always_inline)) {
                        const double2 v = a2[(2 * t < n) ? t : 0];
                        acc[2 * t] = (2 * t < n) ? v.x : 0.0;
                        acc[2 * t + 1] = (2 * t + 1 < n) ? v.y : 0.0;
                    });
                } else {
                    static_for<NMAX>([&](auto c) __attribute__((always_inline)) { acc[c] = (c < n) ? a[c < n ? c : 0] : 0.0; });
                }
            }
            SPROF(3);
            if (direct && tb + TR < mA) {   // this tile lives in registers now: the next one loads during the normalisation
                WSYNC();
                copy_async(tile, A + (size_t)(tb + TR) * n, ((mA - tb - TR) < TR ? (mA - tb - TR) : TR) * n);
            }
            // normalise (utils.c:586-613), d (utils.c:499-544 / 664-676 + 151-159), blocked store
            double s = 0;
            static_for<NMAX>([&](auto c) __attribute__((always_inline)) { if (c < n) s += acc[c] * acc[c]; });
            double scal = 1.0;
            int rowbad = 0;
            if (own) {
                if (s < st.zero_tol) {
                    if (bu_gi < -st.zero_tol || bl_gi > st.zero_tol)
                        if (!(sens[gi] & DAQP_IMMUTABLE) && !(sens[gi] & DAQP_SOFT)) rowbad = 1;
                    sens[gi] = DAQP_IMMUTABLE;
                } else scal = 1 / sqrt(s);
            }
            const bool scale_it = !(s < st.zero_tol);
            static_for<NMAX>([&](auto c) __attribute__((always_inline)) { if (c < n && scale_it) acc[c] *= scal; });
            double dsum = 0;
            if (!unc) static_for<NMAX>([&](auto c) __attribute__((always_inline)) { if (c < n) dsum += acc[c] * vv[c]; });
            if (own) {
                sc[gi] = scal;
                if (unc) {
                    const double u0 = bu_gi - sunc, l0 = bl_gi - sunc;
                    if (u0 < -st.primal_tol || l0 > st.primal_tol) feasible = 0;
                    du[gi] = u0 * scal; dl[gi] = l0 * scal;
                } else {
                    du[gi] = bu_gi * scal + dsum;
                    dl[gi] = bl_gi * scal + dsum;
                }
                double2 *dst = reinterpret_cast<double2 *>(Mq) + ((size_t)(gi >> 6) * b.npair) * 64 + (gi & 63);
                static_for<NMAX / 2>([&](auto t) __attribute__((always_inline)) {
                    if (t < b.npair) {
                        double2 vpair;
                        vpair.x = acc[2 * t];
                        vpair.y = (2 * t + 1 < n) ? acc[2 * t + 1] : 0.0;
                        dst[(size_t)t * 64] = vpair;
                    }
                });
            }
            if (__any(rowbad)) flag = DAQP_EXIT_INFEASIBLE;
            SPROF(4);
        }
    }
    const int all_feasible = __all(feasible);
    int sing = kEmpty;
    if (flag > 0 && unc && all_feasible) { sing = DAQP_UNCONSTRAINED_OPTIMAL; activate = 0; }
    if (flag > 0) {
        if (lane < n) { b.v[(size_t)q * n + lane] = vv[lane]; if (unc) b.xunc[(size_t)q * n + lane] = xu[lane]; }
    }
    for (int i = lane; i < m; i += 64) b.sense[(size_t)q * m + i] = sens[i];
    SPROF(5);
    if (lane == 0) {
        qs->n_active = 0; qs->reuse_ind = 0; qs->sing_ind = sing; qs->iterations = 0;
        qs->lam_swapped = 0; qs->setup_flag = flag; qs->need_activate = (flag > 0) ? activate : 0; qs->pad_ = 0;
        qs->exitflag = flag; qs->fval = 0; qs->soft_slack = 0; qs->diag_h = diag; qs->n_prox = (flag > 0) ? nprox : 0;
        qs->upd_flag = 0;
        if (kProfile && b.prof) for (int i = 0; i < 10; ++i) b.prof[(size_t)q * 32 + i] = pt[i];
    }
#undef SPROF
}

} // namespace daqp_amd
