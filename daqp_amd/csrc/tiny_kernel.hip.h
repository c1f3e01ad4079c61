// tiny_kernel.hip.h -- k_ldp_tiny: daqp_solve for tiny problems, 64/G problems per wavefront (tiny_ldp.hip.h), PERSISTENT waves:
// the grid is one wave per SIMD; the G lanes of a finished problem write its results and take the next problem off the batch's
// counter, so a wave's passes are not bounded by the slowest of its sixteen problems (measured before: 19.3 passes per wave
// against 9.5 iterations per problem).  Same per-problem state in HBM as k_ldp / k_ldp_reg (packed L with its diagonal slots, the
// five working-set vectors, WS, sense, QState), so the kernels are interchangeable (DAQP_AMD_NO_TINY=1 sends these shapes back
// to the one-wave-per-problem register kernel).
#pragma once
#include "tiny_ldp.hip.h"

namespace daqp_amd {

__host__ __device__ inline bool tiny_shape_ok(int n, int m, int cap) { return n <= TNC && m <= TMR && cap <= TCAP; }
#ifndef DAQP_TINY_REFILL
#define DAQP_TINY_REFILL 8      // finished problems of a wave are retired (and replaced) once this many are waiting (measured on C3: 2 -> 2.67 ms, 4 -> 2.35, 6 -> 2.23, 8 -> 2.16, 12 -> 2.20 per 125 000 solves)
#endif

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
    const int n = b.n, m = b.m, cap = b.cap, N = b.N;
    W w;
    w.sm = smem + qw;
    w.sub = sub;
    w.n = n; w.m = m; w.ms = b.ms;
    w.stp = b.st_dev;
    w.dual_tol = b.st.dual_tol; w.sing_tol = b.st.sing_tol; w.pivot_tol = b.st.pivot_tol; w.rho_soft = b.st.rho_soft;
    w.ep = -b.st.primal_tol;
    w.trace = nullptr; w.trace_cap = b.trace_cap; w.trace_len = 0;
    w.na = 0; w.reuse = 0; w.sing = kEmpty; w.fval = 0; w.soft = 0; w.lamsw = 0; w.has_soft = 0;
    w.id0 = w.id1 = w.slw = w.flw = 0; w.slotmask = 0; w.rs = 0;
    w.gdu = w.gdl = w.gsc = nullptr;
    static_for<TNC>([&](auto j) __attribute__((always_inline)) { w.u[j] = 0.0; });
    static_for<RPL>([&](auto k) __attribute__((always_inline)) {
        static_for<TNC>([&](auto j) __attribute__((always_inline)) { if constexpr (!(k < TRI && j < G * k)) w.M[k][j] = 0.0; });
    });
    TSet S;
    S.fbound = 2 * b.st.fval_bound; S.progress_tol = b.st.progress_tol; S.time_limit = b.st.time_limit;
    S.refactor_tol = b.st.refactor_tol; S.primal_tol = b.st.primal_tol; S.tick_s = b.tick_s;
    S.iter_limit = b.st.iter_limit; S.cycle_tol = b.st.cycle_tol; S.mode = mode;
    S.tstart = (b.st.time_limit > 0.0) ? b.tstart : nullptr;
    S.pend = b.tiny_pend;
    TCtl c;
    c.st = TST_EMPTY; c.kind = TK_CSP; c.it = 1; c.flag = 0; c.repaired = 0; c.stall = 0; c.tl_skip = 0;
    c.depth = 0; c.req_add = 1; c.req_id = 0; c.req_r = 0; c.req_sn = 0; c.after = 0;
    c.act_then = 0; c.act_i = 0; c.act_next = 0; c.act_flag = 1; c.best = -1; c.req_lam = 0; c.req_rhs = 0;
    int q = 0;
    bool drained = false;
    // the problems' LDS columns start from zeros: positions beyond a working set are read (and multiplied by exact zeros) by the
    // position loops, which are bounded by the wave's largest working set; what a finished problem leaves behind is finite
    static_for<kTElems / G>([&](auto e) __attribute__((always_inline)) { w.sm[(G * e + sub) * Q] = 0.0; });
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
#ifdef DAQP_TINY_PROF
    w.prof = b.prof != nullptr && mode == 0;
    static_for<12>([&](auto i) __attribute__((always_inline)) { w.pt[i] = 0; });
    w.pt0 = 0;
    const long long t_loop = (long long)__builtin_readcyclecounter();
    long long t_refill = 0, t_retire = 0;
#endif

    for (;;) {
        // ---- retire: results and the persistent iterate of finished problems
        {
            const int ndone = __popcll(__ballot(c.st == TST_DONE)) / G, nidle = __popcll(__ballot(c.st == TST_DONE || c.st == TST_EMPTY)) / G;
            if (ndone > 0 && (ndone >= DAQP_TINY_REFILL || nidle == Q)) {
#ifdef DAQP_TINY_PROF
                const long long tr0 = (long long)__builtin_readcyclecounter();
#endif
                if (c.st == TST_DONE) {
                    QState *qs = b.qs + q;
                    const size_t qfac = qf(b, q);
                    const double *gsc = b.scaling + qfac * m;
                    double *gv = b.vecs + (size_t)q * 5 * cap;
                    int *gws = b.WS + (size_t)q * cap, *gsense = b.sense + (size_t)q * m;
                    const int flag = (mode == 1) ? c.act_flag : c.flag, iters = c.it;
                    if (mode == 1) {
                        if (sub == 0) { qs->need_activate = 0; if (flag < 0) { qs->setup_flag = flag; qs->exitflag = flag; } }
                    } else {
                        // ldp2qp_solution (daqp.c:111-139) + daqp_extract_result (api.c:455-495).  The (now dead) active-row cache
                        // stages the packed R^-1 (n(n+1)/2 <= 78 elements) and, behind it, the m multipliers by constraint.
                        const int qdiag = qs->diag_h;
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
                    // the persistent iterate (lam in buffer A, lam* in buffer B)
                    const double *vq = w.sm + kTV * Q;
                    const int la = w.lamsw ? TV_LB : TV_LA, lb = w.lamsw ? TV_LA : TV_LB;
                    for (int i = sub; i < cap; i += G) {
                        const bool in = i < TCAP;
                        const int ii = in ? i : 0;
                        const double d0 = vq[(TV_D + ii) * Q], d1 = vq[(TV_XL + ii) * Q], d2 = vq[(TV_ZL + ii) * Q], d3 = vq[(la + ii) * Q];
                        double d4 = vq[(lb + ii) * Q];
                        // ldp2qp_solution scales lam* in place (daqp.c:136-138): the stored iterate -- what daqp_extract_result and
                        // the host mirror of work->lam_star read -- holds the scaled multipliers, as the other kernels leave them
                        if (mode != 1 && flag > 0 && i < w.na) d4 *= gsc[tws_id(w, ii)];
                        gv[i] = in ? d0 : 0.0; gv[cap + i] = in ? d1 : 0.0; gv[2 * cap + i] = in ? d2 : 0.0;
                        gv[3 * cap + i] = in ? d3 : 0.0; gv[4 * cap + i] = in ? d4 : 0.0;
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
                        if (w.trace) w.trace[b.trace_cap - 1] = w.trace_len;
                    }
                    c.st = TST_EMPTY;
                }
#ifdef DAQP_TINY_PROF
                t_retire += (long long)__builtin_readcyclecounter() - tr0;
#endif
            }
        }
        // ---- refill: lanes without a problem take the next ones off the batch's counter
        if (!drained && __any(c.st == TST_EMPTY)) {
#ifdef DAQP_TINY_PROF
            const long long tf0 = (long long)__builtin_readcyclecounter();
#endif
            // ONE atomic per trip for the whole wave (a single word serves ~88 fetches per microsecond: one fetch per problem would
            // cost 1.4 ms per 125 000 problems): the lanes that need a problem take consecutive indices
            bool dry = false;
            for (;;) {
                const bool want = c.st == TST_EMPTY && !dry;
                const unsigned long long wm = __ballot(want);
                if (wm == 0) break;
                const int leader = __ffsll((long long)wm) - 1;
                int base = 0;
                if (lane == leader) base = __hip_atomic_fetch_add(b.tiny_counter, __popcll(wm) / G, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                base = __builtin_amdgcn_readlane(base, leader);
                if (want) {
                    const int t = base + __popcll(wm & ((1ull << (lane - sub)) - 1ull)) / G;
                    if (t >= N) { dry = true; continue; }
                    q = t;
                    QState *qs = b.qs + q;
                    const int sflag = qs->setup_flag, need_act = qs->need_activate, uflag = qs->upd_flag;
                    if (mode == 1 && (sflag < 0 || !need_act)) continue;           // nothing to activate
                    if (sflag < 0 || (mode == 0 && uflag < 0)) {                    // setup (api.c:70-78) / update failed: report that, no solve
                        if (sub == 0) { b.exitflag[q] = sflag < 0 ? sflag : uflag; b.iter[q] = 0; if (b.fval) b.fval[q] = 0; if (b.soft) b.soft[q] = 0; }
                        continue;
                    }
                    const int sing0 = qs->sing_ind;
                    if (mode == 0 && sing0 == DAQP_UNCONSTRAINED_OPTIMAL) {         // api.c:40-45: x = unconstrained optimum, no multipliers
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
                        continue;
                    }
                    // ---- a problem to iterate on: row view (this lane's rows of M and their sense bits -> registers)
                    const size_t qfac = qf(b, q);
                    w.gdu = b.dupper + (size_t)q * m; w.gdl = b.dlower + (size_t)q * m; w.gsc = b.scaling + qfac * m;
                    const int *gsense = b.sense + (size_t)q * m;
                    w.trace = b.trace ? b.trace + (size_t)q * b.trace_cap : nullptr; w.trace_len = 0;
                    w.na = qs->n_active; w.reuse = qs->reuse_ind; w.sing = sing0;
                    w.fval = qs->fval; w.soft = qs->soft_slack;
                    w.lamsw = qs->lam_swapped ? 1 : 0;
                    w.id0 = w.id1 = w.slw = w.flw = 0; w.slotmask = 0;
                    if (S.tstart) {     // settings->time_limit: one stamp per problem and daqp_batch_solve (cf. solve_stamp)
                        const unsigned long long ts = __hip_atomic_load(S.tstart + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (ts == 0 && sub == 0) __hip_atomic_store(S.tstart + q, (unsigned long long)__builtin_amdgcn_s_memrealtime(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                    const double2 *msrc = reinterpret_cast<const double2 *>(b.Mblk + qfac * b.nblk * b.npair * 128);
                    const int npair = b.npair;
                    w.rs = 0;
                    int softbits = 0;
                    static_for<RPL>([&](auto k) __attribute__((always_inline)) {
                        const int r = sub + G * k;
                        const bool ok = r < m;
                        const int rr = ok ? r : 0;
                        static_for<TNC / 2>([&](auto tt) __attribute__((always_inline)) {
                            if constexpr (!(k < TRI && 2 * tt + 1 < G * k)) {
                                const bool okt = ok && tt < npair;
                                const double2 v = msrc[(size_t)(okt ? tt : 0) * 64 + rr];
                                w.M[k][2 * tt] = okt ? v.x : 0.0; w.M[k][2 * tt + 1] = okt ? v.y : 0.0;
                            }
                        });
                        const int sn = ok ? (gsense[rr] & 15) : 0;
                        w.rs |= (unsigned long long)sn << (4 * k);
                        softbits |= sn & DAQP_SOFT;
                    });
                    w.has_soft = gor<G>(softbits) ? 1 : 0;
                    static_for<TNC>([&](auto j) __attribute__((always_inline)) { w.u[j] = 0.0; });
                    if (w.na > 0) {     // the stored iterate (warm start, or the working set an activation launch built)
                        const int na = w.na;
                        double *vq = w.sm + kTV * Q;
                        const double *gv = b.vecs + (size_t)q * 5 * cap;
                        const int *gws = b.WS + (size_t)q * cap;
                        for (int i = sub; i < cap && i < TCAP; i += G) {
                            vq[(TV_D + i) * Q] = gv[i]; vq[(TV_XL + i) * Q] = gv[cap + i]; vq[(TV_ZL + i) * Q] = gv[2 * cap + i];
                            vq[(TV_LA + i) * Q] = gv[3 * cap + i]; vq[(TV_LB + i) * Q] = gv[4 * cap + i];
                        }
                        const double *gL = b.L + (size_t)q * b.ltri;
                        for (int i = 1; i < na; ++i)
                            for (int j = sub; j < i; j += G) w.sm[(kTL + tlidx(i, j)) * Q] = gL[tri(i) + j];
                        for (int i = 0; i < na; ++i) {
                            const int id = gws[i];
                            const int sn = gsense[id] & 15;
                            tws_set(w, i, id, i, sn);
                            w.slotmask |= 1u << i;
                            const double rhs = -((sn & DAQP_LOWER) ? w.gdl[id] : w.gdu[id]);
                            if (sub == 0) vq[(TV_RHS + i) * Q] = rhs;
                            double *row = w.sm + (kTR + i * TNC) * Q;
                            for (int tt = sub; tt < TNC / 2; tt += G) {
                                double2 v; v.x = 0.0; v.y = 0.0;
                                if (tt < npair) v = msrc[(size_t)tt * 64 + id];
                                row[(2 * tt) * Q] = v.x; row[(2 * tt + 1) * Q] = v.y;
                            }
                        }
                    }
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    tstart(w, c, S, need_act != 0);
                }
            }
            drained = __any(dry);
#ifdef DAQP_TINY_PROF
            t_refill += (long long)__builtin_readcyclecounter() - tf0;
#endif
        }
        if (!__any(c.st != TST_EMPTY)) break;
        tpass(w, c, S, q);
    }
#ifdef DAQP_TINY_PROF
    if (w.prof && lane == 0) {      // per wave: [0..9] phase cycles, [10] Gram, [11] passes, [12] kernel cycles, [13] refill, [14] retire
        long long *pr = b.prof + (size_t)blockIdx.x * 32;
        static_for<12>([&](auto i) __attribute__((always_inline)) { pr[i] = w.pt[i]; });
        pr[12] = (long long)__builtin_readcyclecounter() - t_loop; pr[13] = t_refill; pr[14] = t_retire;
    }
#endif
}

} // namespace daqp_amd
