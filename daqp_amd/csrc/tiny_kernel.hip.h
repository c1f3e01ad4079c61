// tiny_kernel.hip.h -- k_ldp_tiny: daqp_solve for tiny problems, 64/G problems per wavefront (tiny_ldp.hip.h).  Same per-problem
// state in HBM as k_ldp / k_ldp_reg (packed L with its diagonal slots, the five working-set vectors, WS, sense, QState), so the
// kernels are interchangeable (DAQP_AMD_NO_TINY=1 sends these shapes back to the one-wave-per-problem register kernel).
#pragma once
#include "tiny_ldp.hip.h"

namespace daqp_amd {

__host__ __device__ inline bool tiny_shape_ok(int n, int m, int cap) { return n <= TNC && m <= TMR && cap <= TCAP; }

// mode 0: daqp_solve; 1: only (re)build the working set from the ACTIVE bits (tail of daqp_update_ldp, utils.c:199-211)
template <int G, int TRI, bool FM>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 1))) void k_ldp_tiny(const BatchDev *__restrict__ bp, int mode)
{
    typedef TW<G, TRI, FM> W;
    constexpr int Q = W::Q, RPL = W::RPL;
    const BatchDev &b = *bp;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int lane = lane_id();
    const int sub = lane & (G - 1), qw = lane / G;
    const int q_raw = blockIdx.x * Q + qw;
    const bool valid = q_raw < b.N;
    const int q = valid ? q_raw : b.N - 1;
    const int n = b.n, m = b.m, cap = b.cap;
    QState *qs = b.qs + q;
    const int sflag = qs->setup_flag;
    const int need_act = qs->need_activate;
    const int qdiag = qs->diag_h;
    bool alive = valid, store_state = valid;
    if (mode == 1 && (sflag < 0 || !need_act)) { alive = false; store_state = false; }
    int early = 0;      // a result this launch reports without iterating: 1 = setup flag, 2 = update flag, 3 = unconstrained optimum
    if (alive && sflag < 0) early = 1;
    const int uflag = qs->upd_flag;
    if (alive && !early && mode == 0 && uflag < 0) early = 2;
    W w;
    w.sm = smem + qw;
    w.sub = sub;
    w.n = n; w.m = m; w.ms = b.ms;
    w.stp = b.st_dev;
    w.dual_tol = b.st.dual_tol; w.sing_tol = b.st.sing_tol; w.pivot_tol = b.st.pivot_tol; w.rho_soft = b.st.rho_soft;
    w.trace = (b.trace && valid) ? b.trace + (size_t)q * b.trace_cap : nullptr;
    w.trace_cap = b.trace_cap; w.trace_len = 0;
    w.na = qs->n_active; w.reuse = qs->reuse_ind; w.sing = qs->sing_ind;
    w.fval = qs->fval; w.soft = qs->soft_slack;
    w.lamsw = qs->lam_swapped ? 1 : 0;
    w.id0 = w.id1 = w.slw = w.flw = 0; w.slotmask = 0;
    if (alive && !early && mode == 0 && w.sing == DAQP_UNCONSTRAINED_OPTIMAL) early = 3;
    if (early) { alive = false; store_state = false; }
    const size_t qfac = qf(b, q);
    const double *gdu = b.dupper + (size_t)q * m, *gdl = b.dlower + (size_t)q * m, *gsc = b.scaling + qfac * m;
    int *gsense = b.sense + (size_t)q * m;
    double *gv = b.vecs + (size_t)q * 5 * cap;
    int *gws = b.WS + (size_t)q * cap;
    unsigned long long t_start = __builtin_amdgcn_s_memrealtime();      // (cf. solve_stamp: one stamp per problem and daqp_batch_solve)
    if (b.tstart) {
        const unsigned long long t = __hip_atomic_load(b.tstart + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (t != 0) t_start = t;
        else if (sub == 0 && valid) __hip_atomic_store(b.tstart + q, t_start, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }

    // ---- row view: this lane's rows of M, their bounds, tolerance and sense bits -> registers
    w.gdu = gdu; w.gdl = gdl; w.gsc = gsc; w.ep = -b.st.primal_tol;
    {
        const double2 *msrc = reinterpret_cast<const double2 *>(b.Mblk + qfac * b.nblk * b.npair * 128);
        const int npair = b.npair;
        w.rs = 0;
        int softbits = 0;
        static_for<RPL>([&](auto k) __attribute__((always_inline)) {
            const int r = sub + G * k;
            const bool ok = alive && r < m;
            const int rr = ok ? r : 0;
            static_for<TNC / 2>([&](auto t) __attribute__((always_inline)) {
                if constexpr (!(k < TRI && 2 * t + 1 < G * k)) {
                    const bool okt = ok && t < npair;
                    const double2 v = msrc[(size_t)(okt ? t : 0) * 64 + rr];
                    w.M[k][2 * t] = okt ? v.x : 0.0; w.M[k][2 * t + 1] = okt ? v.y : 0.0;
                }
            });
            const int sn = ok ? (gsense[rr] & 15) : 0;
            w.rs |= (unsigned long long)sn << (4 * k);
            softbits |= sn & DAQP_SOFT;
        });
        w.has_soft = gor<G>(softbits) ? 1 : 0;
    }
    static_for<TNC>([&](auto j) __attribute__((always_inline)) { w.u[j] = 0.0; });
    // ---- working-set view: the stored iterate (warm start, or the working set the activation launch built)
    if (alive && w.na > 0) {
        const int na = w.na;
        double *vq = w.sm + kTV * Q;
        for (int i = sub; i < cap && i < TCAP; i += G) {
            vq[(TV_D + i) * Q] = gv[i]; vq[(TV_XL + i) * Q] = gv[cap + i]; vq[(TV_ZL + i) * Q] = gv[2 * cap + i];
            vq[(TV_LA + i) * Q] = gv[3 * cap + i]; vq[(TV_LB + i) * Q] = gv[4 * cap + i];
        }
        const double *gL = b.L + (size_t)q * b.ltri;
        for (int i = 1; i < na; ++i)
            for (int j = sub; j < i; j += G) w.sm[(kTL + tlidx(i, j)) * Q] = gL[tri(i) + j];
        const double2 *msrc = reinterpret_cast<const double2 *>(b.Mblk + qfac * b.nblk * b.npair * 128);
        for (int i = 0; i < na; ++i) {
            const int id = gws[i];
            const int sn = gsense[id] & 15;
            tws_set(w, i, id, i, sn);
            w.slotmask |= 1u << i;
            const double rhs = -((sn & DAQP_LOWER) ? gdl[id] : gdu[id]);
            if (sub == 0) vq[(TV_RHS + i) * Q] = rhs;
            double *row = w.sm + (kTR + i * TNC) * Q;
            for (int t = sub; t < TNC / 2; t += G) {
                double2 v; v.x = 0.0; v.y = 0.0;
                if (t < b.npair) v = msrc[(size_t)t * 64 + id];
                row[(2 * t) * Q] = v.x; row[(2 * t + 1) * Q] = v.y;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");

    const TinyOut res = trun(w, mode, alive, need_act != 0, b.tiny_pend + (size_t)q * 3 * TCAP, t_start, b.tick_s);
    const int flag = res.flag, iters = res.iterations;

    // ---- results
    if (mode == 1) {
        if (alive && sub == 0) { qs->need_activate = 0; if (flag < 0) { qs->setup_flag = flag; qs->exitflag = flag; } }
    } else if (early == 1 || early == 2) {
        if (sub == 0) { b.exitflag[q] = early == 1 ? sflag : uflag; b.iter[q] = 0; if (b.fval) b.fval[q] = 0; if (b.soft) b.soft[q] = 0; }
    } else if (early == 3) {     // api.c:40-45: x = unconstrained optimum, no multipliers
        const double *xu = b.xunc + (size_t)q * n, *vq = b.v + (size_t)q * n;
        if (b.x) for (int i = sub; i < n; i += G) b.x[(size_t)q * n + i] = xu[i];
        if (b.lam) for (int i = sub; i < m; i += G) b.lam[(size_t)q * m + i] = 0;
        double fv = 0;
        for (int i = 0; i < n; ++i) { const double vi = vq[i]; fv -= vi * vi; }
        fv *= 0.5;
        if (sub == 0) {
            b.exitflag[q] = DAQP_EXIT_OPTIMAL; b.iter[q] = 1;
            if (b.fval) b.fval[q] = fv;
            if (b.soft) b.soft[q] = 0;
            qs->iterations = 1; qs->fval = 0; qs->soft_slack = 0; qs->exitflag = DAQP_EXIT_OPTIMAL;
        }
    } else if (alive) {
        // ldp2qp_solution (daqp.c:111-139) + daqp_extract_result (api.c:455-495).  The (now dead) active-row cache stages the
        // packed R^-1 (n(n+1)/2 <= 78 elements) and, behind it, the m multipliers by constraint.
        const double *Rq = b.Rinv + qfac * b.rtri, *vq = b.v + (size_t)q * n;
        double *Rl = w.sm + kTR * Q, *lamq = w.sm + (kTR + 80) * Q;
        const double *ls = w.sm + TLAMS(w) * Q;
        double lsv[TCAP];
        static_for<TCAP>([&](auto i) __attribute__((always_inline)) { lsv[i] = ls[i * Q]; });
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        if (flag > 0) for (int e = sub; e < b.rtri; e += G) Rl[e * Q] = Rq[e];
        for (int r = sub; r < m; r += G) lamq[r * Q] = 0.0;
        double ux[TNC], fv = w.fval;             // fval - |v|^2 in index order (api.c:471-477)
        static_for<TNC>([&](auto j) __attribute__((always_inline)) {
            const double vj = (j < n) ? vq[j < n ? j : 0] : 0.0;
            ux[j] = (flag > 0) ? w.u[j] - vj : w.u[j];
            fv = msub<FM>(fv, vj, vj);
        });
        fv *= 0.5;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        static_for<TCAP>([&](auto i) __attribute__((always_inline)) {
            if (i < w.na) {
                const int id = tws_id(w, i);
                const double sc = (flag > 0) ? gsc[id] : 1.0;
                if (sub == 0) lamq[id * Q] = (flag > 0) ? lsv[i] * sc : lsv[i];
            }
        });
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        if (b.x) {
            for (int i = sub; i < n; i += G) {
                double xi = 0.0;
                static_for<TNC>([&](auto j) __attribute__((always_inline)) { xi = (j == i) ? ux[j] : xi; });
                if (flag > 0) {
                    const double *row = Rl + roff(i, n) * Q;
                    xi = xi * row[i * Q];
                    static_for<TNC>([&](auto j) __attribute__((always_inline)) {
                        const double t = xi + row[j * Q] * ux[j];
                        xi = (j > i && j < n) ? t : xi;
                    });
                    if (i < b.ms && !qdiag) xi /= gsc[i];      // daqp.c:124-134: no division in the RinvD branch
                }
                b.x[(size_t)q * n + i] = xi;
            }
        }
        if (b.lam) for (int r = sub; r < m; r += G) b.lam[(size_t)q * m + r] = lamq[r * Q];
        if (sub == 0) {
            b.exitflag[q] = flag; b.iter[q] = iters;
            if (b.fval) b.fval[q] = fv;
            if (b.soft) b.soft[q] = w.soft;
            qs->iterations = iters; qs->exitflag = flag; qs->need_activate = 0;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    }
    // ---- store the persistent iterate (lam in buffer A, lam* in buffer B)
    if (store_state) {
        const double *vq = w.sm + kTV * Q;
        const int la = w.lamsw ? TV_LB : TV_LA, lb = w.lamsw ? TV_LA : TV_LB;
        if (mode != 1 && early == 0) {
            // (the epilogue above overwrote the row cache only; the vectors are intact)
        }
        for (int i = sub; i < cap; i += G) {
            const bool in = i < TCAP;
            gv[i] = in ? vq[(TV_D + i) * Q] : 0.0; gv[cap + i] = in ? vq[(TV_XL + i) * Q] : 0.0; gv[2 * cap + i] = in ? vq[(TV_ZL + i) * Q] : 0.0;
            gv[3 * cap + i] = in ? vq[(la + i) * Q] : 0.0; gv[4 * cap + i] = in ? vq[(lb + i) * Q] : 0.0;
            gws[i] = (i < w.na) ? tws_id(w, i) : -1;
        }
        static_for<RPL>([&](auto k) __attribute__((always_inline)) {
            const int r = sub + G * k;
            if (r < m) gsense[r] = (int)((w.rs >> (4 * k)) & 15);
        });
        double *gL = b.L + (size_t)q * b.ltri;
        for (int i = 1; i < w.na; ++i)
            for (int j = sub; j < i; j += G) gL[tri(i) + j] = w.sm[(kTL + tlidx(i, j)) * Q];
        if (sub == 0) {
            qs->n_active = w.na; qs->reuse_ind = w.reuse; qs->sing_ind = w.sing;
            qs->lam_swapped = 0;
            qs->fval = w.fval; qs->soft_slack = w.soft;
            if (b.trace) b.trace[(size_t)q * b.trace_cap + b.trace_cap - 1] = w.trace_len;
        }
    }
}

} // namespace daqp_amd
