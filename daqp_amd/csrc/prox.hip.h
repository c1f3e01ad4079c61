// daqp_amd/csrc/prox.hip.h -- the proximal outer loop for singular (or forcibly regularised) Hessians and for LPs.
//
// Reference: daqp_prox and its gradient_step (src/daqp_prox.c:21-303, QP and LP branches), the regularisation logic of
// daqp_update_Rinv (src/utils.c:223-391) and daqp_get_proximal_regularization (src/utils.c:393-432).
//
// The inner least-distance problems are solved by the very same kernels as every other problem (k_ldp_reg / k_ldp with
// a fused or eager UPDATE_v|UPDATE_d): the outer loop is a handful of O(n) kernels around those launches, driven from the
// host (daqp_amd.hip: solve_with_prox).  A problem that sits out a launch -- an ordinary QP while the proximal ones
// iterate, or a proximal one that has converged -- carries the negative setup flag DAQP_PROX_SKIP, which every kernel
// already treats as "nothing to do"; the flag is put back when the driver is done.
#pragma once
#include "kernels.hip.h"

namespace daqp_amd {

struct ProxDev {
    double *eps;        // [N] shift the factor in use was built with (as reconstructed by utils.c:393-432); an LP: its current smoothing weight
    double *hshift;     // [N] shift tried by the current setup pass
    int *tries;         // [N] doublings so far (utils.c:358)
    double *center;     // [N][n] the reference's work->x between inner solves: survives daqp_batch_solve calls
    double *xold;       // [N][n]
    double *feff;       // [N][n] f - eps * P * center (the linear term of the inner problem)
    int *state;         // [N][4] active, total inner iterations, centre relaxed, outer iterations
    int *saved_flag;    // [N] setup flag of a problem that is sitting out
    int *t_flag, *t_iter;       // [N] outputs of the inner launches
    double *t_fval, *t_soft;    // [N]
    int *counter;       // [4] device-side counts read back by the host
    int lp;             // the batch is an LP batch (H == NULL): R = I, adaptive eps, gradient steps (daqp_prox.c LP branch)
};

// utils.c:13-20
__device__ inline double prox_eps_scaled(const DAQPSettings &st, double hscale)
{
    double eps = st.eps_prox;
    if (eps < 0.0) eps = -eps;
    const double lo = sqrt(st.zero_tol) * hscale;
    if (eps > 0.0 && eps < lo) eps = lo;
    return eps;
}
__device__ inline double hessian_scale(const double *H, int n)
{
    double sc = 0.0;
    for (int i = 0; i < n; ++i) { double a = H[(size_t)i * n + i]; if (a < 0.0) a = -a; if (a > sc) sc = a; }
    return sc;
}

// One thread per problem.  op 0: first shift of every flagged problem (all of them in forced mode), utils.c:276,365-369;
// op 1: after a failed shifted pass -- double the shift, give up after 16 doublings (utils.c:357-360);
// op 2: count the flagged ones only; op 3: flag every problem that is fine so far (LP batches: the one setup pass is the
// generic kernel's).  counter[0] += problems still flagged.
__global__ void k_prox_shift(BatchDev b, ProxDev p, int op)
{
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= b.N) return;
    QState *qs = b.qs + q;
    int flag = qs->setup_flag;
    if (op == 0) {
        if (b.st.eps_prox > 0.0 && flag > 0) { flag = DAQP_NEEDS_SHIFT; qs->setup_flag = flag; }
        if (flag == DAQP_NEEDS_SHIFT) {
            p.hshift[q] = prox_eps_scaled(b.st, hessian_scale(b.H + (size_t)q * b.n * b.n, b.n));
            p.tries[q] = 0;
        }
    } else if (op == 3) {
        if (flag >= 0) { flag = DAQP_NEEDS_SHIFT; qs->setup_flag = flag; }
        p.hshift[q] = 0.0; p.tries[q] = 0;
    } else if (op == 1 && flag == DAQP_NEEDS_SHIFT) {
        const double eps = p.hshift[q];
        if (eps <= 0 || p.tries[q]++ >= 16) { flag = DAQP_EXIT_NONCONVEX; qs->setup_flag = flag; qs->exitflag = flag; }
        else p.hshift[q] = eps * 2.0;
    }
    if (flag == DAQP_NEEDS_SHIFT) atomicAdd(p.counter, 1);
}

// One thread per problem, after the setup passes: the shift of the outer loop as the reference reconstructs it at
// solve time from the factor (utils.c:393-432).  counter[1] += proximal problems.
__global__ void k_prox_final(BatchDev b, ProxDev p)
{
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= b.N) return;
    const QState *qs = b.qs + q;
    double eps = 0.0;
    if (qs->setup_flag > 0 && qs->n_prox > 0 && p.lp) { eps = 1.0; atomicAdd(p.counter + 1, 1); }
    else if (qs->setup_flag > 0 && qs->n_prox > 0) {
        const double *H = b.H + (size_t)q * b.n * b.n;
        eps = prox_eps_scaled(b.st, hessian_scale(H, b.n));
        if (!qs->diag_h && eps > 0.0) {
            double rinv = b.Rinv[(size_t)q * b.rtri];
            if (b.ms > 0) rinv /= b.scaling[(size_t)q * b.m];
            const double recovered = 1.0 / (rinv * rinv) - H[0];
            while (1.5 * eps < recovered) eps *= 2.0;
        }
        atomicAdd(p.counter + 1, 1);
    }
    p.eps[q] = eps;
}

// A shared factorisation that needed the shift (daqp_batch_setup_shared): every problem of the batch iterates with the one eps
__global__ void k_prox_share(BatchDev b, ProxDev p)
{
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= b.N || q == 0) return;
    p.eps[q] = p.eps[0];
}

// One thread per problem: a solve launch that went out BEFORE the host knew of the singular Hessians (daqp_batch_solve straight after
// a setup) reported the internal "needs the shift" code for them.  Once the regularising passes have run, every problem whose setup
// ended in a failure reports its FINAL flag (-5 after the doublings, -1 for a zero row of the shifted pass: utils.c:354-377) exactly as
// a solve launch after the passes would have (api.c:70-78: x / lam untouched, no iterations).
__global__ void k_report_failed_setups(BatchDev b)
{
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= b.N) return;
    const int sflag = b.qs[q].setup_flag;
    if (sflag >= 0 || sflag == DAQP_PROX_SKIP) return;
    b.exitflag[q] = sflag; b.iter[q] = 0;
    if (b.fval) b.fval[q] = 0;
    if (b.soft) b.soft[q] = 0;
}

// One thread per problem: who takes part in the next launches.
//  which 0 (before the ordinary problems are solved): proximal problems start their loop (daqp_prox.c:24-36) and sit out
//  which 1 (before the outer iterations): proximal problems come back, the ordinary ones sit out
//  which 2 (done): everybody gets the own flag back
__global__ void k_prox_mark(BatchDev b, ProxDev p, int which)
{
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= b.N) return;
    QState *qs = b.qs + q;
    int *st = p.state + 4 * (size_t)q;
    if (which == 0) {
        const int prox = qs->setup_flag > 0 && qs->n_prox > 0;
        st[0] = prox; st[1] = 0; st[2] = 0; st[3] = 0;
        if (prox && p.lp) p.eps[q] = 1.0;   // daqp_prox.c:41
        if (prox) { p.saved_flag[q] = qs->setup_flag; qs->setup_flag = DAQP_PROX_SKIP; }
    } else if (which == 1) {
        if (st[0]) qs->setup_flag = p.saved_flag[q];
        else if (qs->setup_flag >= 0) { p.saved_flag[q] = qs->setup_flag; qs->setup_flag = DAQP_PROX_SKIP; }
    } else if (qs->setup_flag == DAQP_PROX_SKIP) qs->setup_flag = p.saved_flag[q];
}

// The input of the next inner problem (daqp_prox.c:66-104,125-126): f - eps*P*x (an LP: eps*f - x with the adapted eps)
// and xold <- x, from the centre.  Whole wave; every lane reads only centre entries it wrote itself (index = lane + 64 j).
// total: inner iterations so far in this solve, last_it: those of the last inner solve.
__device__ inline void prox_next_input(const BatchDev &b, const ProxDev &p, const double *f, int q, int total, int last_it)
{
    const int lane = threadIdx.x, n = b.n;
    if (p.lp) {   // smoothing weight: x10 while the inner LP stalls, x0.9 otherwise, at most 1e3 (daqp_prox.c:69-78)
        double eps = p.eps[q];
        if (total > 0) eps *= (last_it == 1) ? 10.0 : 0.9;
        if (eps > 1e3) eps = 1e3;
        for (int i = lane; i < n; i += 64) {
            const double x = p.center[(size_t)q * n + i];
            p.feff[(size_t)q * n + i] = f[(size_t)q * n + i] * eps - x;
            p.xold[(size_t)q * n + i] = x;
        }
        if (lane == 0) p.eps[q] = eps;
        return;
    }
    const double eps = p.eps[q];
    const int *mask = b.prox_mask + qf(b, q) * n;
    for (int i = lane; i < n; i += 64) {
        const double x = p.center[(size_t)q * n + i];
        p.feff[(size_t)q * n + i] = f[(size_t)q * n + i] - (mask[i] ? eps : 0.0) * x;
        p.xold[(size_t)q * n + i] = x;
    }
}
// the first outer iteration's input (the later ones are formed by k_prox_post / k_lp_gradient as they decide to go on)
__global__ __launch_bounds__(64) void k_prox_pre(BatchDev b, ProxDev p, const double *f)
{
    const int q = blockIdx.x;
    if (!p.state[4 * (size_t)q]) return;
    prox_next_input(b, p, f, q, 0, 0);
}

// The end of a problem's loop (daqp_prox.c:200-221, api.c:455-495): exit flag (the iteration budget overrides), fval,
// duals of an LP back from the smoothed problem, outputs.  x is the final iterate; the whole wave calls this.
struct ProxOut { const double *f; double *lam, *fval, *soft; int *flag, *iter; };
__device__ inline void prox_finish(const BatchDev &b, const ProxDev &p, const ProxOut &o, int q, int flag, int total, const double *x)
{
    const int lane = threadIdx.x, n = b.n, m = b.m;
    QState *qs = b.qs + q;
    const double eps = p.eps[q];
    if (total >= b.st.iter_limit) flag = DAQP_EXIT_ITERLIMIT;
    if (p.lp && o.lam) for (int j = lane; j < m; j += 64) o.lam[(size_t)q * m + j] /= eps;   // daqp_prox.c:203-206
    if (lane == 0) {
        double fv;
        if (p.lp) {   // api.c:479-482
            const double *f = o.f + (size_t)q * n;
            fv = 0;
            for (int i = 0; i < n; ++i) fv += f[i] * x[i];
        } else {      // fval += eps * ||P x||^2, then 1/2 (fval - ||v||^2), summed in index order (daqp_prox.c:208-218, api.c:471-477)
            const int *mask = b.prox_mask + qf(b, q) * n;
            const double *v = b.v + (size_t)q * n;
            double pn = 0.0;
            for (int i = 0; i < n; ++i) if (mask[i]) pn += x[i] * x[i];
            fv = qs->fval + eps * pn;
            qs->fval = fv;
            for (int i = 0; i < n; ++i) fv -= v[i] * v[i];
            fv *= 0.5;
        }
        if (o.fval) o.fval[q] = fv;
        if (o.soft) o.soft[q] = p.t_soft[q];
        o.flag[q] = flag; o.iter[q] = total;
        qs->exitflag = flag; qs->iterations = total;
        p.state[4 * (size_t)q] = 0;
        if (qs->setup_flag >= 0) { p.saved_flag[q] = qs->setup_flag; qs->setup_flag = DAQP_PROX_SKIP; }
    }
}

// One wave per problem, after an inner solve: daqp_prox.c:137-198.  x is the inner solution (already R^-1 (u - v), in the
// output array); the per-problem outputs are written when its loop ends.  state[0] after this: 0 finished, 1 goes on,
// 2 goes on after a gradient step (LP, k_lp_gradient).  counter[2] += problems that go on, counter[3] += those of state 2.
__global__ __launch_bounds__(64) void k_prox_post(BatchDev b, ProxDev p, const double *x_all, ProxOut o)
{
    const int q = blockIdx.x, lane = threadIdx.x, n = b.n;
    int *st = p.state + 4 * (size_t)q;
    if (!st[0]) return;
    QState *qs = b.qs + q;
    const double eps = p.eps[q];
    const double *x = x_all + (size_t)q * n, *xold = p.xold + (size_t)q * n;
    double *center = p.center + (size_t)q * n;
    int flag = p.t_flag[q];
    const int it = p.t_iter[q], limit = b.st.iter_limit;
    const int total = st[1] + it;
    int relaxed = st[2], done = 0, grad = 0;
    double eta = b.st.eta_prox;
    if (eta < 0.0) {   // automatic tolerance (daqp_prox.c:53-58, constants.h:16,22)
        eta = 1e-6;
        if (b.st.dual_tol != 1e-12 && 0.1 * b.st.dual_tol < eta) eta = 0.1 * b.st.dual_tol;
    }
    if (flag < 0) done = 1;                       // the inner solver failed: its flag is the answer
    else if (eps == 0) done = 1;                  // no shift after all: one solve (daqp_prox.c:139)
    else {
        const double tol = p.lp ? eta * eps : eta / eps;   // fixed point ||x - xold||_inf < tol (daqp_prox.c:159-172)
        int moved = 0;
        for (int i = lane; i < n; i += 64) { const double df = x[i] - xold[i]; moved |= (df > tol || df < -tol) ? 1 : 0; }
        moved = __any(moved);
        if (!moved) {
            if (relaxed && total < limit) {       // confirm from the feasible iterate
                relaxed = 0;
                for (int i = lane; i < n; i += 64) center[i] = x[i];
            } else { flag = DAQP_EXIT_OPTIMAL; done = 1; }
        } else if (!p.lp && it == 1 && total < limit) {   // unchanged working set: the proximal map is affine, over-relax it
            for (int i = lane; i < n; i += 64) center[i] = xold[i] + 1.5 * (x[i] - xold[i]);
            relaxed = 1;
        } else {
            relaxed = 0;
            for (int i = lane; i < n; i += 64) center[i] = x[i];
            // an LP iterate that is not at a vertex walks to the next constraint first (daqp_prox.c:187-197)
            if (p.lp && it == 1 && qs->n_active != n) grad = 1;
        }
        if (!done && !grad && total >= limit) done = 1;
    }
    if (lane == 0) { st[1] = total; st[2] = relaxed; st[3] += 1; }
    if (done) {
        if (flag >= 0 || total >= limit) for (int i = lane; i < n; i += 64) center[i] = x[i];
        prox_finish(b, p, o, q, flag, total, x);
    } else {
        if (lane == 0) { st[0] = grad ? 2 : 1; atomicAdd(p.counter + 2, 1); if (grad) atomicAdd(p.counter + 3, 1); }
        if (!grad) prox_next_input(b, p, o.f, q, total, it);
    }
}

// gradient_step (daqp_prox.c:232-303) for the LP problems that asked for it (state 2): move x along x - xold to the first
// blocking constraint and put that constraint into the working set (daqp_add_constraint: LDL' append + pivoting).  One
// wave per problem on the generic wave primitives; the persistent iterate is loaded and stored exactly as k_ldp does.
template <int C, bool SPILL>
__global__ __launch_bounds__(64) void k_lp_gradient(BatchDev b, ProxDev p, double *x_all, ProxOut o)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int q = blockIdx.x, lane = lane_id();
    int *st = p.state + 4 * (size_t)q;
    if (st[0] != 2) return;
    const int n = b.n, m = b.m, ms = b.ms, cap = b.cap;
    QState *qs = b.qs + q;
    const LdpLds lo = ldp_lds(n, m, cap, SPILL);
    int *ibase = reinterpret_cast<int *>(smem + lo.dbl);
    Wave<C, 0, 0> w;
    w.profiling = false;
#pragma unroll
    for (int i = 0; i < 8; ++i) w.prof[i] = 0;
    w.n = n; w.m = m; w.ms = ms; w.cap = cap; w.npair = b.npair; w.nblk = b.nblk; w.ldr = b.ldr;
    if (SPILL) { w.L = b.L + (size_t)q * b.ltri; w.rowc = b.rowc_g + (size_t)q * cap * b.ldr; }
    else { w.L = smem + lo.L; w.rowc = smem + lo.rowc; }
    w.D = smem + lo.D; w.xl = smem + lo.xl; w.zl = smem + lo.zl;
    double *lamA = smem + lo.lamA, *lamB = smem + lo.lamB;
    w.u = smem + lo.u; w.pend_lam = smem + lo.pend_lam;
    w.ws = ibase + lo.ws; w.sense = ibase + lo.sense; w.pend_id = ibase + lo.pend_id;
    w.Mblk = b.Mblk + qf(b, q) * b.nblk * b.npair * 128;
    w.dupper = b.dupper + (size_t)q * m; w.dlower = b.dlower + (size_t)q * m; w.scaling = b.scaling + qf(b, q) * m;
    w.st = b.st;
    w.trace = nullptr; w.trace_cap = 0; w.trace_len = 0;
    w.na = qs->n_active; w.reuse = qs->reuse_ind; w.sing = qs->sing_ind;
    w.fval = qs->fval; w.soft = qs->soft_slack;
    const int swapped = qs->lam_swapped;
    w.lam = swapped ? lamB : lamA; w.lams = swapped ? lamA : lamB;
    int *gsense = b.sense + (size_t)q * m;
    int softbits = 0;
    for (int i = lane; i < m; i += 64) { const int s = gsense[i]; w.sense[i] = s; softbits |= s & DAQP_SOFT; }
    w.has_soft = __any(softbits) ? 1 : 0;
    double *gv = b.vecs + (size_t)q * 5 * cap;
    int *gws = b.WS + (size_t)q * cap;
    for (int i = lane; i < cap; i += 64) {
        w.D[i] = gv[i]; w.xl[i] = gv[cap + i]; w.zl[i] = gv[2 * cap + i];
        lamA[i] = gv[3 * cap + i]; lamB[i] = gv[4 * cap + i];
        w.ws[i] = gws[i];
    }
    if (!SPILL) {
        const int used = tri(w.na);
        const double *gL = b.L + (size_t)q * b.ltri;
        for (int e = lane; e < used; e += 64) w.L[e] = gL[e];
    }
    for (int e = lane; e < round_up(n > 64 ? n : 64, 2) + 2; e += 64) w.u[e] = 0;
    WSYNC();
    for (int i = 0; i < w.na; ++i) fetch_row(w, w.ws[i], i);
    WSYNC();

    // ---- the first blocking constraint along dx = x - xold: rows in index order, strict comparisons against the running
    // minimum exactly as the reference's loop (the row products in parallel, the selection as a uniform scan over the lanes)
    const double *x = x_all + (size_t)q * n, *xold = p.xold + (size_t)q * n;
    const double *bu = b.bu + (size_t)q * m, *bl = b.bl + (size_t)q * m;
    int pick = kEmpty, lower = 0;
    double amin = DAQP_INF;
    for (int base = 0; base < m; base += 64) {
        const int j = base + lane;
        double ax = 0, ds = 0, up = 0, lo_ = 0;
        int skip = 1;
        if (j < m) {
            skip = (w.sense[j] & (DAQP_ACTIVE + DAQP_IMMUTABLE)) ? 1 : 0;
            up = bu[j]; lo_ = bl[j];
            if (j < ms) { ax = x[j]; ds = x[j] - xold[j]; }
            else if (!skip) {
                const double2 *row = reinterpret_cast<const double2 *>(w.Mblk) + ((size_t)(j >> 6) * w.npair) * 64 + (j & 63);
                for (int t = 0; t < w.npair; ++t) {
                    const double2 mv = row[(size_t)t * 64];
                    ax += mv.x * x[2 * t]; ds -= mv.x * xold[2 * t];
                    if (2 * t + 1 < n) { ax += mv.y * x[2 * t + 1]; ds -= mv.y * xold[2 * t + 1]; }
                }
                ds += ax;
                ax /= w.scaling[j]; ds /= w.scaling[j];
            }
        }
        const int cnt = (m - base < 64) ? m - base : 64;
        for (int l = 0; l < cnt; ++l) {
            if (__builtin_amdgcn_readlane(skip, l)) continue;
            const double ax_l = rl(ax, l), ds_l = rl(ds, l), up_l = rl(up, l), lo_l = rl(lo_, l);
            if (ds_l > 0 && up_l < DAQP_INF && up_l - ax_l < ds_l * amin) {
                pick = base + l; lower = 0; amin = (up_l - ax_l) / ds_l;
            } else if (ds_l < 0 && lo_l > -DAQP_INF && lo_l - ax_l > ds_l * amin) {
                pick = base + l; lower = 1; amin = (lo_l - ax_l) / ds_l;
            }
        }
    }
    const int total = st[1];
    double *center = p.center + (size_t)q * n;
    double *xo = x_all + (size_t)q * n;
    if (pick == kEmpty) {                       // nothing blocks: the LP is unbounded along -f
        prox_finish(b, p, o, q, DAQP_EXIT_UNBOUNDED, total, x);
        return;
    }
    for (int k = lane; k < n; k += 64) { const double xn = x[k] + amin * (x[k] - xold[k]); center[k] = xn; xo[k] = xn; }
    if (lane == 0) { if (lower) w.sense[pick] |= DAQP_LOWER; else w.sense[pick] &= ~DAQP_LOWER; }
    WSYNC();
    add_constraint(w, pick, lower ? -1.0 : 1.0);
    // ---- store the persistent iterate
    for (int i = lane; i < cap; i += 64) {
        gv[i] = w.D[i]; gv[cap + i] = w.xl[i]; gv[2 * cap + i] = w.zl[i];
        gv[3 * cap + i] = lamA[i]; gv[4 * cap + i] = lamB[i];
        gws[i] = (i < w.na) ? w.ws[i] : -1;
    }
    for (int i = lane; i < m; i += 64) gsense[i] = w.sense[i];
    if (!SPILL) {
        const int used = tri(w.na);
        double *gL = b.L + (size_t)q * b.ltri;
        for (int e = lane; e < used; e += 64) gL[e] = w.L[e];
    }
    if (lane == 0) {
        qs->n_active = w.na; qs->reuse_ind = w.reuse; qs->sing_ind = w.sing;
        qs->lam_swapped = (w.lam == lamB) ? 1 : 0;
        st[0] = 1;
    }
    __threadfence();
    if (total >= b.st.iter_limit) prox_finish(b, p, o, q, DAQP_EXIT_ITERLIMIT, total, xo);   // the budget ran out on this very step
    else prox_next_input(b, p, o.f, q, total, 1);   // (a gradient step follows an inner solve of one iteration)
}

} // namespace daqp_amd
