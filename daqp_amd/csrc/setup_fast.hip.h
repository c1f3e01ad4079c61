// setup_fast.hip.h -- QP -> LDP for n <= NMAX <= 64, one wavefront per QP, restructured around
// the one thing that matters at one-to-three waves per SIMD: never wait for LDS inside a
// dependency chain.  Arithmetic (operation ORDER included) is exactly that of k_setup /
// reference src/utils.c:223-687; only the schedule differs:
//   * Cholesky row i (utils.c:335-352): lane <-> column j; the k-ordered chain
//       r_ij -= r_ki * r_kj   runs on register operands loaded eight steps ahead;
//   * R -> R^-1 (utils.c:380-389): lane <-> row k, LEFT-looking: t_j = (r_kj*(-1/r_kk)
//       - sum_{i<j} r_ij t_i) / r_jj with the row's own partial results in a conflict-free
//       per-lane LDS column (no read-modify-write of shared rows);
//   * M = A R^-1 (utils.c:434-472): lane <-> row of A, r-OUTER: all n accumulators of the row
//       live in registers and every step  m_c += rinv_rc * a_r  (c >= r) is independent of its
//       neighbours, so the wave runs at VALU rate instead of LDS latency.  Per output the
//       terms still arrive diagonal first, then decreasing r.
#pragma once
#include "wave_ldp_reg.hip.h"
#include "kernels.hip.h"

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

// dst[r*ld + c] = src[r*n + c] for r < rows, c < n: every lane walks the contiguous source with stride 64,
// keeps (row, col) incrementally (no division) and has 8 loads in flight before the first LDS store --
// at <= 3 waves per CU nothing else would hide the HBM latency of a load-store-load-store loop.
__device__ __forceinline__ void stage_rows(double *dst, const double *src, int rows, int n, int ld)
{
    const int lane = lane_id(), total = rows * n;
    int r = 0, c = lane;
    while (c >= n) { c -= n; ++r; }
    for (int e0 = lane; e0 < total; e0 += 64 * 8) {
        double v[8];
        int off[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int e = e0 + 64 * q;
            v[q] = (e < total) ? src[e] : 0.0;
            off[q] = r * ld + c;
            c += 64;
            while (c >= n) { c -= n; ++r; }
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) if (e0 + 64 * q < total) dst[off[q]] = v[q];
    }
}

// LDS: [ Rsq: n x nsq zero-padded square R^-1, overlaying the packed Cholesky factor once that is
// dead | f v xu | scaling dupper dlower | tile (64 x ldr: inverse scratch, then the A tiles) | sense ]
struct FastLds { int R, fv, vv, xu, sc, du, dl, tile, sens, total_bytes; };
__host__ __device__ inline FastLds fast_lds(int n, int m)
{
    FastLds s;
    const int nsq = round_up(n, 8), rt = round_up(n * (n + 1) / 2, 2), np = round_up(n, 2), mp = round_up(m, 2), ldr = n | 1;
    int o = 0;
    s.R = o; o += (n * nsq > rt ? n * nsq : rt) + 8;
    s.fv = o; o += np; s.vv = o; o += np; s.xu = o; o += np;
    s.sc = o; o += mp; s.du = o; o += mp; s.dl = o; o += mp;
    s.tile = o; o += round_up(64 * ldr, 2);
    s.sens = o;
    s.total_bytes = o * 8 + round_up(m, 4) * 4;
    return s;
}

template <int NMAX>
__global__ __launch_bounds__(64) void k_setup_fast(BatchDev b, int mask)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int q = blockIdx.x, lane = lane_id();
    const int n = b.n, m = b.m, ms = b.ms, mA = b.mA, ldr = b.ldr, nsq = round_up(n, 8);
    const FastLds o = fast_lds(n, m);
    double *R = smem + o.R, *Rsq = smem + o.R, *fl = smem + o.fv, *vv = smem + o.vv, *xu = smem + o.xu;
    double *sc = smem + o.sc, *du = smem + o.du, *dl = smem + o.dl, *tile = smem + o.tile;
    int *sens = reinterpret_cast<int *>(smem + o.sens);
    const double *H = b.H + (size_t)q * n * n, *f = b.f + (size_t)q * n, *A = b.A + (size_t)q * mA * n;
    const double *bu = b.bu + (size_t)q * m, *bl = b.bl + (size_t)q * m;
    const DAQPSettings &st = b.st;
    QState *qs = b.qs + q;
    int flag = 1, activate = 0;
    long long pt[6] = {0, 0, 0, 0, 0, 0};
    long long t0 = b.prof ? (long long)__builtin_readcyclecounter() : 0;
#define SPROF(slot) do { if (b.prof) { const long long t1 = (long long)__builtin_readcyclecounter(); pt[slot] += t1 - t0; t0 = t1; } } while (0)

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
    if (st.eps_prox > 0.0) flag = DAQP_EXIT_UNSUPPORTED;
    if (lane < n) fl[lane] = f[lane];
    // pack 1/2 (H + H') (utils.c:318-324): H is staged through the (still unused) tile area
    if (flag > 0) {
        stage_rows(tile, H, n, n, ldr);
        WSYNC();
        for (int i = 0; i < n; ++i) {
            const int j = i + lane;
            if (j < n) R[roff(i, n) + j] = (j == i) ? tile[i * ldr + i] : 0.5 * (tile[i * ldr + j] + tile[j * ldr + i]);
        }
    }
    WSYNC();

    // --- Cholesky (utils.c:335-352)
    double pmin = DAQP_INF, pmax = 0.0;
    if (flag > 0) {
        for (int i = 0; i < n && flag > 0; ++i) {
            const int pi = roff(i, n);
            const int j = i + lane;
            const bool own = j < n;
            const int jj = own ? j : i;                 // clamp: inactive lanes recompute the diagonal chain
            double acc = R[pi + jj];
            acc = chain_sub8(acc, i, [&](int k) { return R[roff(k, n) + i]; }, [&](int k) { return R[roff(k, n) + jj]; });
            const double dg = rl(acc, 0);
            if (dg <= st.zero_tol) { flag = (st.eps_prox == 0.0) ? DAQP_EXIT_NONCONVEX : DAQP_EXIT_UNSUPPORTED; break; }
            if (dg < pmin) pmin = dg;
            if (dg > pmax) pmax = dg;
            const double dgi = 1 / sqrt(dg);
            if (own) R[pi + j] = (lane == 0) ? dgi : acc * dgi;
            WSYNC();
        }
        if (flag > 0 && pmin <= st.zero_tol * pmax)
            flag = (st.eps_prox == 0.0) ? DAQP_EXIT_NONCONVEX : DAQP_EXIT_UNSUPPORTED;
    }
    SPROF(0);
    // --- R -> R^-1 (utils.c:380-389), lane <-> row k.  T[i*64 + lane] = t_i of this lane's row (0 for i <= k),
    // kept in the tile area; the finished row is written to Ro.
    double *T = tile;
    if (flag > 0) {
        const int k = lane;
        const bool own = k < n;
        const int pk = own ? roff(k, n) : 0;
        const double rkk = own ? R[pk + k] : 0.0;
        T[lane] = 0.0;   // i = 0 is never a t_i of any row (i > k >= 0)
        for (int j = 1; j < n; ++j) {
            const int jc = (own && j > k) ? j : (own ? k : 0);          // in-range address for idle lanes
            double acc = R[pk + jc] * -rkk;                             // r_kj *= -(1/r_kk)
            acc = chain_sub8(acc, j - 1, [&](int ii) { return R[roff(ii + 1, n) + j]; },
                             [&](int ii) { return T[(ii + 1) * 64 + lane]; });
            const double tj = acc * R[roff(j, n) + j];                   // t_j *= 1/r_jj
            T[j * 64 + lane] = (own && j > k) ? tj : 0.0;
        }
        WSYNC();
        // the Cholesky factor is dead: expand R^-1 into the zero-padded square Rsq[r*nsq + c]
        for (int e = lane; e < n * nsq; e += 64) Rsq[e] = 0.0;
        WSYNC();
        if (own) {
            Rsq[k * nsq + k] = rkk;
            for (int j = k + 1; j < n; ++j) Rsq[k * nsq + j] = T[j * 64 + lane];
        }
        WSYNC();
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

    // --- general rows, 64 at a time through the LDS tile; lane <-> row
    int feasible = 1;
    double *Mq = b.Mblk + (size_t)q * b.nblk * b.npair * 128;
    if (flag > 0) {
        for (int tb = 0; tb < mA && flag > 0; tb += 64) {
            const int rows = (mA - tb) < 64 ? (mA - tb) : 64;
            WSYNC();
            stage_rows(tile, A + (size_t)tb * n, rows, n, ldr);
            WSYNC();
            const int k = tb + lane;
            const bool own = lane < rows;
            const double *a = tile + (own ? lane : 0) * ldr;
            const int gi = ms + (own ? k : tb);
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
                // (upper-triangular: K only up to the column tile's last column) 16-column tiles of R^-1.
                // A: lane l supplies A[l&15][l>>4], B: R^-1[l>>4][l&15], D: col = l&15, row = (l>>4) + 4*reg.
                // The summation order differs from the reference's (fp64, fused): M agrees to ~1e-16 relative.
                typedef double v4d __attribute__((ext_vector_type(4)));
                const int lr = lane & 15, lk = lane >> 4, ktn = (n + 3) >> 2;
                for (int rt = 0; rt * 16 < rows; ++rt) {
                    v4d acc4[4];
                    static_for<4>([&](auto ct) __attribute__((always_inline)) { acc4[ct] = (v4d){0.0, 0.0, 0.0, 0.0}; });
                    const int arow = rt * 16 + lr;
                    const bool rowok = arow < rows;
                    const double *arowp = tile + (rowok ? arow : 0) * ldr;
                    for (int kt = 0; kt < ktn; ++kt) {
                        const int kk = 4 * kt + lk;
                        const bool kok = kk < n;
                        const double aload = arowp[kok ? kk : 0];
                        const double av = (rowok && kok) ? aload : 0.0;
                        static_for<4>([&](auto ct) __attribute__((always_inline)) {
                            if (16 * ct < n && kt <= 4 * ct + 3) {
                                const int cc = 16 * ct + lr;
                                const bool ok = kok && cc < n;
                                const double bload = Rsq[(kok ? kk : 0) * nsq + (cc < n ? cc : 0)];
                                acc4[ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, ok ? bload : 0.0, acc4[ct], 0, 0, 0);
                            }
                        });
                    }
                    WSYNC();   // every lane's reads of this row tile precede the in-place overwrite below
                    static_for<4>([&](auto ct) __attribute__((always_inline)) {
                        static_for<4>([&](auto r) __attribute__((always_inline)) {
                            const int row = rt * 16 + lk + 4 * r, col = 16 * ct + lr;
                            if (col < n && row < rows) tile[row * ldr + col] = acc4[ct][(int)r];
                        });
                    });
                }
                WSYNC();
                static_for<NMAX>([&](auto c) __attribute__((always_inline)) { acc[c] = (c < n) ? a[c < n ? c : 0] : 0.0; });
            }
            SPROF(3);
            // normalise (utils.c:586-613), d (utils.c:499-544 / 664-676 + 151-159), blocked store
            double s = 0;
            static_for<NMAX>([&](auto c) __attribute__((always_inline)) { if (c < n) s += acc[c] * acc[c]; });
            double scal = 1.0;
            int rowbad = 0;
            if (own) {
                if (s < st.zero_tol) {
                    if (bu[gi] < -st.zero_tol || bl[gi] > st.zero_tol)
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
                    const double u0 = bu[gi] - sunc, l0 = bl[gi] - sunc;
                    if (u0 < -st.primal_tol || l0 > st.primal_tol) feasible = 0;
                    du[gi] = u0 * scal; dl[gi] = l0 * scal;
                } else {
                    du[gi] = bu[gi] * scal + dsum;
                    dl[gi] = bl[gi] * scal + dsum;
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
    // --- simple bounds: normalise rows < ms of R^-1 (utils.c:569-585), their d, their dense image in M
    if (flag > 0) {
        WSYNC();
        if (lane < ms) {
            const int i = lane;
            double *Ri = Rsq + i * nsq;
            double s = 0;
            for (int j = i; j < n; ++j) s += Ri[j] * Ri[j];
            s = 1 / sqrt(s);
            sc[i] = s;
            for (int j = i; j < n; ++j) Ri[j] *= s;
            if (unc) {
                const double u0 = bu[i] - xu[i], l0 = bl[i] - xu[i];
                if (u0 < -st.primal_tol || l0 > st.primal_tol) feasible = 0;
                du[i] = u0 * s; dl[i] = l0 * s;
            } else {
                double t = 0;
                for (int j = i; j < n; ++j) t += Ri[j] * vv[j];
                du[i] = bu[i] * s + t;
                dl[i] = bl[i] * s + t;
            }
            double2 *dst = reinterpret_cast<double2 *>(Mq) + ((size_t)(i >> 6) * b.npair) * 64 + (i & 63);
            for (int t = 0; t < b.npair; ++t) {
                double2 vpair;
                vpair.x = (2 * t >= i) ? Ri[2 * t] : 0.0;
                vpair.y = (2 * t + 1 >= i && 2 * t + 1 < n) ? Ri[2 * t + 1] : 0.0;
                dst[(size_t)t * 64] = vpair;
            }
        }
        WSYNC();
    }
    const int all_feasible = __all(feasible);
    int sing = kEmpty;
    if (flag > 0 && unc && all_feasible) { sing = DAQP_UNCONSTRAINED_OPTIMAL; activate = 0; }
    if (flag > 0) {
        for (int e = lane; e < n * n; e += 64) {   // packed upper image for the solve kernel / warm updates
            const int i = e / n, j = e - i * n;
            if (j >= i) b.Rinv[(size_t)q * b.rtri + roff(i, n) + j] = Rsq[i * nsq + j];
        }
        if (lane < n) { b.v[(size_t)q * n + lane] = vv[lane]; if (unc) b.xunc[(size_t)q * n + lane] = xu[lane]; }
        for (int i = lane; i < m; i += 64) {
            b.scaling[(size_t)q * m + i] = sc[i];
            b.dupper[(size_t)q * m + i] = du[i];
            b.dlower[(size_t)q * m + i] = dl[i];
        }
    }
    for (int i = lane; i < m; i += 64) b.sense[(size_t)q * m + i] = sens[i];
    SPROF(5);
    if (lane == 0) {
        qs->n_active = 0; qs->reuse_ind = 0; qs->sing_ind = sing; qs->iterations = 0;
        qs->lam_swapped = 0; qs->setup_flag = flag; qs->need_activate = (flag > 0) ? activate : 0;
        qs->exitflag = flag; qs->fval = 0; qs->soft_slack = 0;
        if (b.prof) for (int i = 0; i < 6; ++i) b.prof[(size_t)q * 32 + i] = pt[i];
    }
#undef SPROF
}

} // namespace daqp_amd
