// wg_kernel.hip.h -- k_ldp_wg: daqp_solve for large problems, one workgroup of W <= WMAX wavefronts per QP, persistent
// workgroups (wg_ldp.hip.h has the design).  Same global state layout as k_ldp, so the two are interchangeable: a problem
// whose working set outgrows the LDS-resident L is flagged in b.fallback and solved by k_ldp<C, true> right after.
#pragma once
#include "batch_dev.hip.h"
#include "wg_ldp.hip.h"

namespace daqp_amd {

// mode 0: daqp_solve; mode 1: only (re)build the working set from the ACTIVE bits (tail of daqp_update_ldp);
// mode | 4: only the problems flagged in b.fallback -- this kernel behind its own tiered launch (TIER below), and the ONE-WAVE kernel (k_ldp) behind both
// EX: the arithmetic mode as a compile-time constant (b.exact_setup decides which instantiation is launched).  Every function below
// is inlined into the kernel and branches on c.exact: as a run-time field both modes' code -- the reference's ordered chains AND the
// inverse factor with its tree sums -- shared one register allocation, and the default mode's launch carried the chains' live ranges
// (488 bytes of scratch per lane, 169 spilled registers in k_ldp_wg<4> at round 4).
// TIER: the cold solves of the default arithmetic at TWO four-wave workgroups per CU (one problem's serial master phases under the other's bandwidth
// phases): rows < wg_r0 of the inverse factor in LDS, the rest in the problem's slot of the stored factor (wg_ldp.hip.h: WROW).  Takes only problems
// that start from an empty working set without soft rows; everything else -- and whatever leaves the inverse-factor representation beyond wg_r0 rows
// -- is flagged in b.fallback for the launch behind it (the same kernel without tiers, mode | 4).
// mode | 4 (no TIER): only the problems flagged in b.fallback.
template <int C, bool EX, bool TIER = false>
__global__ __launch_bounds__(64 * kWgMaxWaves) void k_ldp_wg(BatchDev b, int mode_in)
{
    static_assert(!(TIER && EX), "the tiers hold the inverse factor: default arithmetic only");
    const int mode = mode_in & 3;
    const bool flagged_only = !TIER && (mode_in & 4) != 0;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    __shared__ int q_sh;
    __shared__ int m_int[8];       // master -> everybody after the iteration: flag, iterations, na, reuse, sing, lam swapped, overflow
    __shared__ double m_dbl[2];    // fval, soft
    const int wv = wg_wave();   // (wave-uniform by construction: say so, or the master's whole body sits in a "divergent" branch)
    const int n = b.n, m = b.m, cap = b.cap, W = (int)(blockDim.x >> 6);
    WgCtx c;
    const int capL = TIER ? b.wg_r0 : b.wg_capL;
    c.n = n; c.m = m; c.ms = b.ms; c.cap = cap; c.capL = capL; c.npair = b.npair; c.nblk = b.nblk; c.ldr = wg_row_stride(n); c.capT = b.wg_capT;
    c.W = W; c.exact = EX ? 1 : 0;
    c.oL = wg_lds_L(C, m); c.lmax = wg_round_up(capL * (capL + 1) / 2, 2) - 1;
    c.tier = TIER ? 1 : 0; c.r0 = capL; c.capW = TIER ? cap : capL; c.gL = nullptr;
    c.rowc = b.wg_rowc + (size_t)blockIdx.x * cap * wg_row_stride(n);
    c.rowcT = b.wg_rowcT + (size_t)blockIdx.x * n * b.wg_capT;
    const int T = (int)blockDim.x;

    for (;;) {
        // (thread coordinates taken anew, behind an opaque asm, in each part of the loop body: whatever is computed from a kernel-wide
        //  `tid` -- LDS and HBM addresses of the load / write-back loops below -- is invariant in this loop, gets hoisted out of it and
        //  then sits in registers across the master's state machine and the workers' command loop: ~40 spilled registers of 267)
        const int tid = wg_tid(), lane = tid & 63;
        __syncthreads();                       // the previous problem is completely done with LDS
        if (tid == 0) q_sh = atomicAdd(b.wg_counter, 1);
        __syncthreads();
        const int q = uni(q_sh);
        if (q >= b.N) break;
        if (flagged_only && !uni(b.fallback[q])) continue;
        QState *qs = b.qs + q;
        const int sflag = uni(qs->setup_flag), need_act = uni(qs->need_activate), uflag = uni(qs->upd_flag);
        if (mode == 1) { if (sflag < 0 || !need_act) continue; }
        if (sflag < 0) {   // setup failed: x/lam untouched, no solve (api.c:70-78)
            if (tid == 0) { b.exitflag[q] = sflag; b.iter[q] = 0; if (b.fval) b.fval[q] = 0; if (b.soft) b.soft[q] = 0; b.fallback[q] = 0; }
            continue;
        }
        if (mode == 0 && uflag < 0) {   // the last update failed its bound check: report that, keep the state (see k_update)
            if (tid == 0) { b.exitflag[q] = uflag; b.iter[q] = 0; if (b.fval) b.fval[q] = 0; if (b.soft) b.soft[q] = 0; b.fallback[q] = 0; }
            continue;
        }
        const int sing0 = uni(qs->sing_ind);
        if (sing0 == DAQP_UNCONSTRAINED_OPTIMAL && mode == 0) {   // api.c:40-45: x = unconstrained optimum, no multipliers
            const double *xu = b.xunc + (size_t)q * n, *vq = b.v + (size_t)q * n;
            if (b.x) for (int i = tid; i < n; i += T) b.x[(size_t)q * n + i] = xu[i];
            if (b.lam) for (int i = tid; i < m; i += T) b.lam[(size_t)q * m + i] = 0;
            if (tid == 0) {
                double fv = 0;
                for (int i = 0; i < n; ++i) { const double vi = vq[i]; fv -= vi * vi; }
                fv *= 0.5;
                b.exitflag[q] = DAQP_EXIT_OPTIMAL; b.iter[q] = 1;
                if (b.fval) b.fval[q] = fv;
                if (b.soft) b.soft[q] = 0;
                qs->iterations = 1; qs->fval = 0; qs->soft_slack = 0; qs->exitflag = DAQP_EXIT_OPTIMAL;
                b.fallback[q] = 0;
            }
            continue;
        }
        const int na0 = uni(qs->n_active);
        if (TIER && (na0 != 0 || need_act || mode != 0)) {      // not a cold start: the launch behind this one
            if (tid == 0) b.fallback[q] = 1;
            continue;
        }
        if (na0 > c.capL) {        // a warm start that does not fit: the one-wave kernel takes the problem as it is
            if (tid == 0) b.fallback[q] = 1;
            continue;
        }
        const size_t qfac = qf(b, q);
        c.Mblk = b.Mblk + qfac * b.nblk * b.npair * 128;
        c.M32 = b.M32 ? b.M32 + qfac * b.nblk * b.nquad * 256 : nullptr; c.nquad = b.nquad;
        c.dupper = b.dupper + (size_t)q * m; c.dlower = b.dlower + (size_t)q * m; c.scaling = b.scaling + qfac * m;
        // ---- load the persistent iterate
        int *gsense = b.sense + (size_t)q * m;
        int softbits = 0;
        for (int i = tid; i < m; i += T) { const int s = gsense[i]; SI(c, sense)[i] = s; softbits |= s & DAQP_SOFT; }
        const int has_soft = uni(__syncthreads_or(softbits) ? 1 : 0);
        if (TIER && has_soft) {
            if (tid == 0) b.fallback[q] = 1;
            continue;
        }
        c.gL = b.L + (size_t)q * b.ltri;
        double *gv = b.vecs + (size_t)q * 5 * cap;
        int *gws = b.WS + (size_t)q * cap;
        for (int i = tid; i < cap; i += T) {
            SD(c, D)[i] = gv[i]; SD(c, xl)[i] = gv[cap + i]; SD(c, zl)[i] = gv[2 * cap + i];
            SD(c, lamA)[i] = gv[3 * cap + i]; SD(c, lamB)[i] = gv[4 * cap + i];
            SI(c, ws)[i] = gws[i];
            SI(c, slot)[i] = i;
            SI(c, freestk)[i] = cap - 1 - i;             // slots 0 .. na0-1 are taken: the stack holds cap-1 ... na0 (top = lowest free)
        }
        {
            const int used = tri(na0);
            const double *gL = b.L + (size_t)q * b.ltri;
            for (int e = tid; e < used; e += T) SDL(c)[e] = gL[e];
        }
        for (int e = tid; e < wg_round_up(n, 2) + 2; e += T) SD(c, u)[e] = 0;
        __syncthreads();
        for (int i = wv; i < na0; i += W) {         // rebuild the active-row scratch, a row per wave and trip
            const int id = SI(c, ws)[i];
            if (lane == 0) SI(c, slot_id)[i] = id;
            const double2 *src = reinterpret_cast<const double2 *>(c.Mblk) + ((size_t)(id >> 6) * c.npair) * 64 + (id & 63);
            for (int t = lane; t < c.npair; t += 64) {
                const double2 v = src[(size_t)t * 64];
                wg_store_row_pair(c, i, t, v, 2 * t + 1 < n);
            }
        }
        __syncthreads();

        if (wv == 0) {
            WgWave<C> w;
            w.c = c;
            w.t_start = solve_stamp(b.tstart, q); w.tick_s = b.tick_s;
            w.profiling = (b.prof != nullptr) && mode == 0;
            if (w.profiling && lane < 24) reinterpret_cast<long long *>(wg_sm() + WgL<C>::prof)[lane] = 0;
            w.stp = b.st_dev;
            w.trace = b.trace ? b.trace + (size_t)q * b.trace_cap : nullptr;
            w.trace_cap = b.trace_cap; w.trace_len = 0;
            w.na = na0; w.reuse = uni(qs->reuse_ind); w.sing = sing0;
            w.fval = und(qs->fval); w.soft = und(qs->soft_slack);
            w.has_soft = has_soft;
            w.nfree = cap - na0; w.hi_slot = na0 - 1; w.overflow = 0;
            w.lam_b = uni(qs->lam_swapped) ? 1 : 0;
            // inverse-factor representation: default arithmetic, cold start, no soft rows (wg_ldp.hip.h)
            w.use_w = (b.wg_inverse && !c.exact && mode == 0 && na0 == 0 && !need_act && !has_soft) ? 1 : 0;
            w.fast_na = -1;
            int iters = 0;
            const int flag = wrun(w, mode, need_act != 0, iters);
            if (!w.overflow) wleave_w(w, w.na, true);                    // the stored iterate is always L   // (need_activate at mode 0: defensive, setup/update runs mode 1 itself)
            if (lane == 0) {
                m_int[0] = flag; m_int[1] = iters; m_int[2] = w.na; m_int[3] = w.reuse; m_int[4] = w.sing;
                m_int[5] = w.lam_b; m_int[6] = w.overflow; m_int[7] = w.trace_len;
                m_dbl[0] = w.fval; m_dbl[1] = w.soft;
                SI(c, cmd)[0] = WG_EXIT;
                if (w.profiling) {
                    for (int i = 0; i < 16; ++i) b.prof[(size_t)q * 32 + i] = reinterpret_cast<long long *>(wg_sm() + WgL<C>::prof)[i];
                    for (int i = 16; i < 23; ++i) b.prof[(size_t)q * 32 + 9 + i] = reinterpret_cast<long long *>(wg_sm() + WgL<C>::prof)[i];   // [25..28]: scans, fp64 re-scans; [29..31]: inside an append
                }
            }
            __syncthreads();
        } else wg_serve<C>(c, b.st.primal_tol);
        __syncthreads();

        // ---- everybody: outputs and the persistent iterate
        const int tide = wg_tid();             // (see the top of the loop)
        const int flag = uni(m_int[0]), iters = uni(m_int[1]), na = uni(m_int[2]);
        if (uni(m_int[6])) {            // overflow: nothing of the problem's state in HBM has been touched; the one-wave kernel redoes it
            if (tide == 0) b.fallback[q] = 1;
            continue;
        }
        double *lams = uni(m_int[5]) ? SD(c, lamA) : SD(c, lamB);
        if (mode == 1) {
            if (tide == 0) { qs->need_activate = 0; if (flag < 0) { qs->setup_flag = flag; qs->exitflag = flag; } }
        } else {
            // ldp2qp_solution (daqp.c:111-139) + daqp_extract_result (api.c:455-495)
            const double *Rq = b.Rinv + qfac * b.rtri, *vq = b.v + (size_t)q * n;
            double *vl = SD(c, mnew);                                    // v staged in LDS (the new-row buffer is free now)
            for (int i = tide; i < n; i += T) vl[i] = vq[i];
            __syncthreads();
            if (flag > 0) {
                for (int i = tide; i < n; i += T) SD(c, u)[i] = SD(c, u)[i] - vl[i];
                for (int i = tide; i < na; i += T) lams[i] *= c.scaling[SI(c, ws)[i]];
            }
            __syncthreads();
            if (flag > 0) {
                const int diag = qs->diag_h;
                for (int i = tide; i < n; i += T) {
                    const double *row = Rq + roff(i, n);
                    double xi = SD(c, u)[i] * row[i];
                    for (int j = i + 1; j < n; ++j) xi += row[j] * SD(c, u)[j];
                    if (i < b.ms && !diag) xi /= c.scaling[i];   // daqp.c:124-134: no division in the RinvD branch
                    if (b.x) b.x[(size_t)q * n + i] = xi;
                }
            } else if (b.x) {
                for (int i = tide; i < n; i += T) b.x[(size_t)q * n + i] = SD(c, u)[i];
            }
            if (b.lam) {
                for (int i = tide; i < m; i += T) b.lam[(size_t)q * m + i] = 0;
                __syncthreads();
                for (int i = tide; i < na; i += T) b.lam[(size_t)q * m + SI(c, ws)[i]] = lams[i];
            }
            if (tide == 0) {
                double fv = m_dbl[0];
                for (int i = 0; i < n; ++i) { const double vi = vl[i]; fv -= vi * vi; }
                fv *= 0.5;
                b.exitflag[q] = flag; b.iter[q] = iters;
                if (b.fval) b.fval[q] = fv;
                if (b.soft) b.soft[q] = m_dbl[1];
                qs->iterations = iters; qs->exitflag = flag; qs->need_activate = 0;
            }
        }
        for (int i = tide; i < cap; i += T) {
            gv[i] = SD(c, D)[i]; gv[cap + i] = SD(c, xl)[i]; gv[2 * cap + i] = SD(c, zl)[i];
            gv[3 * cap + i] = SD(c, lamA)[i]; gv[4 * cap + i] = SD(c, lamB)[i];
            gws[i] = (i < na) ? SI(c, ws)[i] : -1;
        }
        for (int i = tide; i < m; i += T) gsense[i] = SI(c, sense)[i];
        {
            const int used = tri((TIER && na > c.r0) ? c.r0 : na);       // (TIER: the rows from r0 on are in place already)
            double *gL = b.L + (size_t)q * b.ltri;
            for (int e = tide; e < used; e += T) gL[e] = SDL(c)[e];
        }
        if (tide == 0) {
            qs->n_active = na; qs->reuse_ind = m_int[3]; qs->sing_ind = m_int[4];
            qs->lam_swapped = m_int[5];
            qs->fval = m_dbl[0]; qs->soft_slack = m_dbl[1];
            if (b.trace) b.trace[(size_t)q * b.trace_cap + b.trace_cap - 1] = m_int[7];
            b.fallback[q] = 0;
        }
    }
}

} // namespace daqp_amd
