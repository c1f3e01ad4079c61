// tiny_setup.hip.h -- QP -> LDP for tiny problems (n <= 12, m <= 48), SIXTEEN problems per wavefront: G = 4 lanes per problem.
// Arithmetic, operation order included, is that of the reference (src/utils.c:58-687) and therefore of k_setup_fast in its exact
// mode -- in BOTH arithmetic modes of the library: at this size the matrix cores have nothing to offer (one 16-row MFMA tile would
// hold a single problem's M) and the whole setup is ~4 k flops per problem.  What changes is the schedule:
//   * the 12 x 12 factorisation (symmetrisation, Cholesky with 1/r_ii on the diagonal, R -> R^-1 in place, v = R^-T f, the
//     unconstrained optimum) runs REPLICATED in the four lanes of a problem on a register-resident packed triangle with
//     compile-time indices -- no cross-lane traffic, no LDS; problems with n < 12 are padded with an identity block, which the
//     recurrences leave untouched;
//   * lane s then forms the rows s, s+4, ... of the LDP: for a general row M_r = A_r R^-1 (diagonal term first, then decreasing
//     row index: utils.c:441-453), its normalisation and its d; for a simple bound the normalised row of R^-1 (selected out of
//     the replicated triangle);
//   * one wave's loads of A are 64 x 96 contiguous bytes per row slot, its stores of the blocked M image 64 B per problem and
//     pair of columns.
// One k_setup_fast wave spent ~4.7 k instructions on ONE such problem (lanes 12..63 idle through the factorisation); here
// ~5 k instructions serve sixteen.
// Singular / indefinite Hessians and forced regularisation leave with DAQP_NEEDS_SHIFT exactly as in k_setup_fast: the host's
// regularising re-run (k_setup_fast<16, true>) takes those problems.
#pragma once
#include "wave_ldp_reg.hip.h"
#include "batch_dev.hip.h"

namespace daqp_amd {

constexpr int TNC = 12;     // columns
constexpr int TMR = 48;     // constraint rows
// cross-lane traffic inside a group of four lanes: DPP quad permutes
template <int CTRL> __device__ __forceinline__ int dpp_i(int v) { return __builtin_amdgcn_update_dpp(v, v, CTRL, 0xF, 0xF, false); }
template <int G> __device__ __forceinline__ int gor(int v) { static_assert(G == 4, "groups of four lanes"); v |= dpp_i<0xB1>(v); v |= dpp_i<0x4E>(v); return v; }

template <int G>
__global__ __launch_bounds__(64) void k_setup_tiny(BatchDev b, int mask)
{
    constexpr int Q = 64 / G, RPL = TMR / G, NC = TNC;
    const int lane = lane_id(), sub = lane & (G - 1), qw = lane / G;
    const int q_raw = blockIdx.x * Q + qw;
    const bool valid = q_raw < b.N;
    const int q = valid ? q_raw : b.N - 1;
    const int n = b.n, m = b.m, ms = b.ms, mA = b.mA;
    const DAQPSettings &st = b.st;
    const double *H = b.H + (size_t)q * n * n, *f = b.f + (size_t)q * n, *A = (b.A != nullptr && mA > 0) ? b.A + (size_t)q * mA * n : H;
    // (the row slots load unconditionally and mask the result: without general rows / without constraints the loads go to the problem's own H)
    const double *bu = m > 0 ? b.bu + (size_t)q * m : H, *bl = m > 0 ? b.bl + (size_t)q * m : H;
    double *sc = b.scaling + (size_t)q * m, *du = b.dupper + (size_t)q * m, *dl = b.dlower + (size_t)q * m;
    QState *qs = b.qs + q;
    const bool force = st.eps_prox > 0.0;

    // --- sense (utils.c:84-91) and early bound check (utils.c:546-567): this lane's rows
    int sens[RPL];
    double bur[RPL], blr[RPL];
    int bad = 0, fixed = 0;
    static_for<RPL>([&](auto k) __attribute__((always_inline)) {
        const int r = sub + G * k;
        const bool ok = r < m;
        const int rr = ok ? r : 0;
        int s = (b.sense_in && ok) ? b.sense_in[(size_t)q * m + rr] : 0;
        bur[k] = bu[rr]; blr[k] = bl[rr];
        if (ok) {
            if (s & DAQP_BINARY) bad |= 2;
            if (!(s & DAQP_IMMUTABLE)) {
                const double diff = bur[k] - blr[k];
                if (diff < -st.primal_tol) bad |= 1;
                else if (diff < st.zero_tol && !(s & DAQP_SOFT)) { s |= DAQP_ACTIVE + DAQP_IMMUTABLE; bad |= 4; }
            }
            fixed |= s & (DAQP_ACTIVE + DAQP_IMMUTABLE);
        }
        sens[k] = s;
    });
    bad = gor<G>(bad);
    fixed = gor<G>(fixed);
    int flag = 1, activate = 0;
    if (b.sense_in) activate = 1;
    if (bad & 4) activate = 1;
    if (bad & 2) flag = DAQP_EXIT_UNSUPPORTED;
    else if (bad & 1) flag = DAQP_EXIT_INFEASIBLE;
    if (force && flag > 0) flag = DAQP_NEEDS_SHIFT;   // forced proximal mode: the host starts with the shifted pass

    // --- 1/2 (H + H') packed upper, padded to 12 x 12 with an identity block (utils.c:318-324); diagonal test (utils.c:245-252)
    double R[NC][NC];      // only i <= j is ever touched
    int offd = 0;
    static_for<NC>([&](auto i) __attribute__((always_inline)) {
        static_for<NC>([&](auto j) __attribute__((always_inline)) {
            if constexpr (i <= j) {
                const bool in = i < n && j < n;
                const double hij = H[in ? i * n + j : 0], hji = H[in ? j * n + i : 0];
                if constexpr (i == j) R[i][j] = in ? hij : 1.0;
                else {
                    R[i][j] = in ? 0.5 * (hij + hji) : 0.0;
                    if (in && (hij > st.zero_tol || hij < -st.zero_tol)) offd = 1;
                }
            }
        });
    });
    const bool isdiag = !offd;
    // (a diagonal H: entries above the diagonal within zero_tol count as zeros -- the RinvD branch never looks at them)
    static_for<NC>([&](auto i) __attribute__((always_inline)) {
        static_for<NC>([&](auto j) __attribute__((always_inline)) { if constexpr (i < j) R[i][j] = isdiag ? 0.0 : R[i][j]; });
    });
    double pmin = DAQP_INF, pmax = 0.0;
    double hsq[NC];        // sqrt(H_ii) in the RinvD branch (utils.c:245-312)
    static_for<NC>([&](auto i) __attribute__((always_inline)) { hsq[i] = 1.0; });
    if (flag > 0 && isdiag) {
        double hscale = 0.0;
        static_for<NC>([&](auto i) __attribute__((always_inline)) { if (i < n) { const double a = R[i][i] < 0 ? -R[i][i] : R[i][i]; if (a > hscale) hscale = a; } });
        const double ftol = hscale > 0 ? st.zero_tol * hscale : st.zero_tol;
        static_for<NC>([&](auto i) __attribute__((always_inline)) {
            if (i < n && flag > 0) {
                const double hd = R[i][i];
                if (hd <= ftol) flag = (st.eps_prox == 0.0 && hd <= st.zero_tol) ? DAQP_EXIT_NONCONVEX : DAQP_NEEDS_SHIFT;   // the reference stops at the first such i
                else { hsq[i] = sqrt(hd); R[i][i] = 1 / hsq[i]; }
            }
        });
    }
    // --- in-place Cholesky, 1/r_ii on the diagonal (utils.c:335-352): row i from the rows above it, subtractions in ascending k
    if (flag > 0 && !isdiag) {
        static_for<NC>([&](auto iv) __attribute__((always_inline)) {
            constexpr int i = decltype(iv)::value;
            double dg = R[i][i];
            static_for<i>([&](auto k) __attribute__((always_inline)) { dg -= R[k][i] * R[k][i]; });
            if (i < n) {
                if (dg <= st.zero_tol && flag > 0) flag = (st.eps_prox == 0.0) ? DAQP_EXIT_NONCONVEX : DAQP_NEEDS_SHIFT;
                if (dg < pmin) pmin = dg;
                if (dg > pmax) pmax = dg;
            }
            dg = 1 / sqrt(dg);
            static_for<NC - 1 - i>([&](auto jj) __attribute__((always_inline)) {
                constexpr int j = i + 1 + jj;
                double t = R[i][j];
                static_for<i>([&](auto k) __attribute__((always_inline)) { t -= R[k][i] * R[k][j]; });
                R[i][j] = t * dg;
            });
            R[i][i] = dg;
        });
        if (flag > 0 && pmin <= st.zero_tol * pmax) flag = (st.eps_prox == 0.0) ? DAQP_EXIT_NONCONVEX : DAQP_NEEDS_SHIFT;   // utils.c:354-356
        // --- R -> R^-1 in place, row by row (utils.c:380-389)
        static_for<NC>([&](auto kv) __attribute__((always_inline)) {
            constexpr int k = decltype(kv)::value;
            static_for<NC - 1 - k>([&](auto jj) __attribute__((always_inline)) { R[k][k + 1 + jj] *= -R[k][k]; });
            static_for<NC - 1 - k>([&](auto ii) __attribute__((always_inline)) {
                constexpr int i = k + 1 + ii;
                R[k][i] *= R[i][i];
                static_for<NC - 1 - i>([&](auto jj) __attribute__((always_inline)) { R[k][i + 1 + jj] -= R[i][i + 1 + jj] * R[k][i]; });
            });
        });
    }
    // --- v = R^-T f (utils.c:474-497): v_i = R_ii f_i, then += R_ji f_j for j = i-1 .. 0
    double fv[NC], v[NC], xu[NC];
    static_for<NC>([&](auto i) __attribute__((always_inline)) { fv[i] = (i < n) ? f[i < n ? i : 0] : 0.0; });
    static_for<NC>([&](auto iv) __attribute__((always_inline)) {
        constexpr int i = decltype(iv)::value;
        double acc = R[i][i] * fv[i];
        static_for<i>([&](auto t) __attribute__((always_inline)) { constexpr int j = i - 1 - t; acc += R[j][i] * fv[j]; });
        v[i] = acc;
    });
    // --- unconstrained optimum (utils.c:618-662)
    const bool unc = flag > 0 && (mask & DAQP_UPDATE_unconstrained) && !fixed;
    static_for<NC>([&](auto iv) __attribute__((always_inline)) {
        constexpr int i = decltype(iv)::value;
        double s = 0.0;
        static_for<NC - i>([&](auto t) __attribute__((always_inline)) { s += R[i][i + t] * v[i + t]; });
        xu[i] = -s;
    });
    if (flag > 0 && (mask & DAQP_UPDATE_eliminate)) {   // eq_elim.c:127-164: that variant is not built
        int neq = 0;
        static_for<RPL>([&](auto k) __attribute__((always_inline)) {
            const int r = sub + G * k;
            if (r >= ms && r < m && ((sens[k] & (DAQP_ACTIVE + DAQP_IMMUTABLE + DAQP_SOFT + DAQP_BINARY)) == (DAQP_ACTIVE + DAQP_IMMUTABLE))) neq++;
        });
        neq += dpp_i<0xB1>(neq); neq += dpp_i<0x4E>(neq);
        if (neq > 5 && 10 * neq > n) flag = DAQP_EXIT_UNSUPPORTED;
    }

    // --- this lane's rows of the LDP
    int feasible = 1, rowbad = 0;
    const bool okq = valid && flag > 0;
    double2 *Mq = reinterpret_cast<double2 *>(b.Mblk + (size_t)q * b.nblk * b.npair * 128);
    double *Rp = b.Rinv + (size_t)q * b.rtri;
    const int npair = b.npair;
    // rows of R^-1 (all of them go to the packed image; those below ms are normalised first: utils.c:569-585)
    static_for<NC / G>([&](auto k) __attribute__((always_inline)) {
        const int i = sub + G * k;
        double row[NC];
        static_for<NC>([&](auto j) __attribute__((always_inline)) {
            double val = 0.0;
            static_for<G>([&](auto s) __attribute__((always_inline)) { if constexpr (G * k + s <= j) val = (sub == s) ? R[G * k + s][j] : val; });
            row[j] = val;      // R^-1[i][j] (0 left of the diagonal)
        });
        double hs = 1.0;
        static_for<G>([&](auto s) __attribute__((always_inline)) { hs = (sub == s) ? hsq[G * k + s] : hs; });
        double s2 = 0.0, scal = 1.0;
        const bool simple = i < ms;
        if (isdiag) scal = hs;              // scaling_i = sqrt(H_ii) (utils.c:309); the row of R^-1 counts as the unit vector
        else {
            static_for<NC>([&](auto j) __attribute__((always_inline)) { const double t = s2 + row[j] * row[j]; s2 = (j >= i) ? t : s2; });
            scal = 1 / sqrt(s2);
        }
        if (okq && i < n) {
            if (simple) {
                double u0, l0;
                if (unc) {
                    double xi = 0.0;
                    static_for<G>([&](auto s) __attribute__((always_inline)) { xi = (sub == s) ? xu[G * k + s] : xi; });
                    u0 = bur[k] - xi; l0 = blr[k] - xi;
                    if (u0 < -st.primal_tol || l0 > st.primal_tol) feasible = 0;
                    u0 *= scal; l0 *= scal;
                } else {
                    double t = 0.0;
                    if (isdiag) { static_for<G>([&](auto s) __attribute__((always_inline)) { t = (sub == s) ? v[G * k + s] : t; }); }   // utils.c:527-531
                    else static_for<NC>([&](auto j) __attribute__((always_inline)) { const double tt = t + (row[j] * scal) * v[j]; t = (j >= i) ? tt : t; });
                    u0 = bur[k] * scal + t; l0 = blr[k] * scal + t;
                }
                sc[i] = scal; du[i] = u0; dl[i] = l0;
                static_for<NC / 2>([&](auto t) __attribute__((always_inline)) {
                    if (t < npair) {
                        double2 vp;
                        if (isdiag) { vp.x = (2 * t == i) ? 1.0 : 0.0; vp.y = (2 * t + 1 == i) ? 1.0 : 0.0; }
                        else { vp.x = (2 * t >= i) ? row[2 * t] * scal : 0.0; vp.y = (2 * t + 1 >= i && 2 * t + 1 < n) ? row[2 * t + 1] * scal : 0.0; }
                        Mq[(size_t)t * 64 + i] = vp;
                    }
                });
            }
            const double rs = (simple && !isdiag) ? scal : 1.0;
            static_for<NC>([&](auto j) __attribute__((always_inline)) { if (j >= i && j < n) Rp[roff(i, n) + j] = row[j] * rs; });
        }
    });
    // general rows: M_r = A_r R^-1, normalisation (utils.c:586-613), d (utils.c:499-544 / 664-676 + 151-159)
    static_for<RPL>([&](auto k) __attribute__((always_inline)) {
        const int r = sub + G * k;
        const bool gen = r >= ms && r < m;
        const double *arow = A + (size_t)(gen ? r - ms : 0) * n;
        double a[NC];
        static_for<NC>([&](auto j) __attribute__((always_inline)) { const double t = arow[j < n ? j : 0]; a[j] = (gen && j < n) ? t : 0.0; });
        double sunc = 0.0;
        static_for<NC>([&](auto j) __attribute__((always_inline)) { sunc += a[j] * xu[j]; });
        double acc[NC];
        static_for<NC>([&](auto cv) __attribute__((always_inline)) {      // column c: diagonal term first, then decreasing row index
            constexpr int c = decltype(cv)::value;
            double t = R[c][c] * a[c];
            static_for<c>([&](auto tt) __attribute__((always_inline)) { constexpr int rr = c - 1 - tt; t += R[rr][c] * a[rr]; });
            acc[c] = t;
        });
        double s = 0.0;
        static_for<NC>([&](auto c) __attribute__((always_inline)) { s += acc[c] * acc[c]; });
        double scal = 1.0;
        const bool zero_row = s < st.zero_tol;
        if (gen && zero_row) {
            if (bur[k] < -st.zero_tol || blr[k] > st.zero_tol)
                if (!(sens[k] & DAQP_IMMUTABLE) && !(sens[k] & DAQP_SOFT)) rowbad = 1;
            sens[k] = DAQP_IMMUTABLE;
        }
        if (!zero_row) scal = 1 / sqrt(s);
        static_for<NC>([&](auto c) __attribute__((always_inline)) { if (!zero_row) acc[c] *= scal; });
        double dsum = 0.0;
        static_for<NC>([&](auto c) __attribute__((always_inline)) { dsum += acc[c] * v[c]; });
        if (okq && gen) {
            sc[r] = scal;
            if (unc) {
                const double u0 = bur[k] - sunc, l0 = blr[k] - sunc;
                if (u0 < -st.primal_tol || l0 > st.primal_tol) feasible = 0;
                du[r] = u0 * scal; dl[r] = l0 * scal;
            } else { du[r] = bur[k] * scal + dsum; dl[r] = blr[k] * scal + dsum; }
            static_for<NC / 2>([&](auto t) __attribute__((always_inline)) {
                if (t < npair) { double2 vp; vp.x = acc[2 * t]; vp.y = acc[2 * t + 1]; Mq[(size_t)t * 64 + r] = vp; }
            });
        }
    });
    if (gor<G>(rowbad) && flag > 0) flag = DAQP_EXIT_INFEASIBLE;
    const int all_feasible = !gor<G>(feasible ? 0 : 1);
    int sing = kEmpty;
    if (flag > 0 && unc && all_feasible) { sing = DAQP_UNCONSTRAINED_OPTIMAL; activate = 0; }
    if (!valid) return;
    if (flag > 0) {
        static_for<NC / G>([&](auto k) __attribute__((always_inline)) {
            const int i = sub + G * k;
            double vi = 0.0, xi = 0.0;
            static_for<G>([&](auto s) __attribute__((always_inline)) { vi = (sub == s) ? v[G * k + s] : vi; xi = (sub == s) ? xu[G * k + s] : xi; });
            if (i < n) { b.v[(size_t)q * n + i] = vi; if (unc) b.xunc[(size_t)q * n + i] = xi; }
        });
    }
    static_for<RPL>([&](auto k) __attribute__((always_inline)) { const int r = sub + G * k; if (r < m) b.sense[(size_t)q * m + r] = sens[k]; });
    if (sub == 0) {
        qs->n_active = 0; qs->reuse_ind = 0; qs->sing_ind = sing; qs->iterations = 0;
        qs->lam_swapped = 0; qs->setup_flag = flag; qs->need_activate = (flag > 0) ? activate : 0; qs->pad_ = 0;
        qs->exitflag = flag; qs->fval = 0; qs->soft_slack = 0; qs->diag_h = (flag > 0 && isdiag) ? 1 : 0; qs->n_prox = 0;
        qs->upd_flag = 0;
    }
}

} // namespace daqp_amd
