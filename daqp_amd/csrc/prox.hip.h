// daqp_amd/csrc/prox.hip.h -- the proximal outer loop for singular (or forcibly regularised) Hessians.
//
// Reference: daqp_prox (src/daqp_prox.c:21-221, QP branch), the regularisation logic of daqp_update_Rinv
// (src/utils.c:223-391) and daqp_get_proximal_regularization (src/utils.c:393-432).
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
    double *eps;        // [N] shift the factor in use was built with (as reconstructed by utils.c:393-432)
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
// op 2: count the flagged ones only.  counter[0] += problems still flagged.
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
    if (qs->setup_flag > 0 && qs->n_prox > 0) {
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
        if (prox) { p.saved_flag[q] = qs->setup_flag; qs->setup_flag = DAQP_PROX_SKIP; }
    } else if (which == 1) {
        if (st[0]) qs->setup_flag = p.saved_flag[q];
        else if (qs->setup_flag >= 0) { p.saved_flag[q] = qs->setup_flag; qs->setup_flag = DAQP_PROX_SKIP; }
    } else if (qs->setup_flag == DAQP_PROX_SKIP) qs->setup_flag = p.saved_flag[q];
}

// One wave per problem: v's input f - eps*P*x and xold <- x (daqp_prox.c:66-104,125-126)
__global__ __launch_bounds__(64) void k_prox_pre(BatchDev b, ProxDev p, const double *f)
{
    const int q = blockIdx.x, lane = threadIdx.x, n = b.n;
    if (!p.state[4 * (size_t)q]) return;
    const double eps = p.eps[q];
    const int *mask = b.prox_mask + (size_t)q * n;
    for (int i = lane; i < n; i += 64) {
        const double x = p.center[(size_t)q * n + i];
        p.feff[(size_t)q * n + i] = f[(size_t)q * n + i] - (mask[i] ? eps : 0.0) * x;
        p.xold[(size_t)q * n + i] = x;
    }
}

// One wave per problem, after an inner solve: daqp_prox.c:137-221.  x is the inner solution (already R^-1 (u - v), in the
// output array), the per-problem outputs are written when its loop ends.  counter[2] += problems that go on.
__global__ __launch_bounds__(64) void k_prox_post(BatchDev b, ProxDev p, const double *x_all, double *o_fval, double *o_soft,
                                                  int *o_flag, int *o_iter)
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
    int relaxed = st[2], done = 0;
    double eta = b.st.eta_prox;
    if (eta < 0.0) {   // automatic tolerance (daqp_prox.c:53-58, constants.h:16,22)
        eta = 1e-6;
        if (b.st.dual_tol != 1e-12 && 0.1 * b.st.dual_tol < eta) eta = 0.1 * b.st.dual_tol;
    }
    if (flag < 0) done = 1;                       // the inner solver failed: its flag is the answer
    else if (eps == 0) done = 1;                  // no shift after all: one solve (daqp_prox.c:139)
    else {
        const double tol = eta / eps;             // fixed point ||x - xold||_inf < tol (daqp_prox.c:159-172)
        int moved = 0;
        for (int i = lane; i < n; i += 64) { const double df = x[i] - xold[i]; moved |= (df > tol || df < -tol) ? 1 : 0; }
        moved = __any(moved);
        if (!moved) {
            if (relaxed && total < limit) {       // confirm from the feasible iterate
                relaxed = 0;
                for (int i = lane; i < n; i += 64) center[i] = x[i];
            } else { flag = DAQP_EXIT_OPTIMAL; done = 1; }
        } else if (it == 1 && total < limit) {    // unchanged working set: the proximal map is affine, over-relax it
            for (int i = lane; i < n; i += 64) center[i] = xold[i] + 1.5 * (x[i] - xold[i]);
            relaxed = 1;
        } else {
            relaxed = 0;
            for (int i = lane; i < n; i += 64) center[i] = x[i];
        }
        if (!done && total >= limit) done = 1;
    }
    if (done) {
        if (flag >= 0 || total >= limit) for (int i = lane; i < n; i += 64) center[i] = x[i];
        if (total >= limit) flag = DAQP_EXIT_ITERLIMIT;
        if (lane == 0) {
            // daqp_prox.c:203-218 then api.c:471-477: fval += eps * ||P x||^2; 1/2 (fval - ||v||^2), summed in index order
            const int *mask = b.prox_mask + (size_t)q * n;
            const double *v = b.v + (size_t)q * n;
            double pn = 0.0;
            for (int i = 0; i < n; ++i) if (mask[i]) pn += x[i] * x[i];
            double fv = qs->fval + eps * pn;
            qs->fval = fv;
            for (int i = 0; i < n; ++i) fv -= v[i] * v[i];
            fv *= 0.5;
            if (o_fval) o_fval[q] = fv;
            if (o_soft) o_soft[q] = p.t_soft[q];
            o_flag[q] = flag; o_iter[q] = total;
            qs->exitflag = flag; qs->iterations = total;
            st[0] = 0;
            if (qs->setup_flag >= 0) { p.saved_flag[q] = qs->setup_flag; qs->setup_flag = DAQP_PROX_SKIP; }
        }
    } else if (lane == 0) atomicAdd(p.counter + 2, 1);
    if (lane == 0) { st[1] = total; st[2] = relaxed; st[3] += 1; }
}

} // namespace daqp_amd
