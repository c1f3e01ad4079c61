// recheck.hip.h -- default arithmetic only: problems that a solve straight after a setup declares INFEASIBLE are set up and solved
// again in the reference's own arithmetic, and that second result is the one reported.
//
// Why: "infeasible" is decided in the singular branch of the iteration (daqp.c:86-93) by comparing components of a singular
// direction with dual_tol = 1e-12 (auxiliary.c:284-287).  For a certificate those components are zero in exact arithmetic: what is
// compared is rounding noise, and the default mode's noise (fused multiply-adds, M = A R^-1 on the matrix cores) is not the
// reference's -- the certificate can come one removal earlier or later (same exit flag, same working set, iter +-1: 2 of 400
// threshold-sitting problems, profiles/r03_degenerate_fast_mode.json).  Infeasible problems are rare and end early, so they are the
// ones worth a second, bit-identical pass: their exit flag, iteration count, multipliers and stored iterate are then the reference's.
//
// Mechanics (host side: recheck_infeasible in daqp_amd.hip): k_mark_infeasible compacts the indices and publishes the count into a
// mapped host word that the host polls; if it is not zero the inputs of those problems are gathered into a small companion batch that runs in the exact
// mode (k_gather_problems), set up and solved there by the ordinary kernels, and results, LDP and iterate are copied back over
// the problem's slots (k_scatter_problems).  No kernel of the hot path knows about any of this.
#pragma once
#include "batch_dev.hip.h"

namespace daqp_amd {

// one thread per problem: solve-time INFEASIBLE of an ordinary problem (its setup succeeded, it is not in the proximal loop --
// that loop already runs the reference's arithmetic in both modes).  The block that finishes last publishes the count straight
// into host memory (a mapped, pinned word the host is polling: it learns the answer a couple of microseconds after this kernel
// ends, without a stream synchronisation -- on a 2.4 ms step of tiny problems the synchronisation cost 2 % of the throughput).
// tick[0]: count, tick[1]: blocks done (both zeroed by the host before the launch).
__global__ void k_mark_infeasible(BatchDev b, int *list, int *tick, int *host_count)
{
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q < b.N) {
        const QState *qs = b.qs + q;
        if (b.exitflag[q] == DAQP_EXIT_INFEASIBLE && qs->setup_flag > 0 && qs->n_prox == 0 && qs->upd_flag >= 0)
            list[atomicAdd(tick, 1)] = q;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        if (atomicAdd(tick + 1, 1) == (int)gridDim.x - 1) {
            __threadfence();
            const int cnt = __hip_atomic_load(tick, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(host_count, cnt, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

template <typename T>
__device__ __forceinline__ void copy_span(T *dst, const T *src, size_t count)
{
    for (size_t e = threadIdx.x; e < count; e += blockDim.x) dst[e] = src[e];
}

// one workgroup per slot of the companion batch: slot i takes problem list[i] (slots beyond `count` repeat the first: the companion
// has a fixed capacity and solves them too, nobody reads their results)
__global__ void k_gather_problems(BatchDev src, const int *list, const int *count, double *H, double *f, double *A, double *bu, double *bl, int *sense)
{
    const int i = blockIdx.x, c = *count;
    const size_t q = (size_t)list[i < c ? i : 0], n = src.n, m = src.m, mA = src.mA;
    copy_span(H + (size_t)i * n * n, src.H + q * n * n, n * n);
    copy_span(f + (size_t)i * n, src.f + q * n, n);
    if (mA) copy_span(A + (size_t)i * mA * n, src.A + q * mA * n, mA * n);
    copy_span(bu + (size_t)i * m, src.bu + q * m, m);
    copy_span(bl + (size_t)i * m, src.bl + q * m, m);
    if (sense && src.sense_in) copy_span(sense + (size_t)i * m, src.sense_in + q * m, m);
}

// one workgroup per re-solved problem: slot i of the companion (`r`) -> problem list[i] of the batch (`b`): outputs, the LDP, the
// stored iterate, the record -- everything a later warm solve, mirror or working-set query reads.  Same shape => same layouts.
__global__ void k_scatter_problems(BatchDev b, BatchDev r, const int *list, const int *count)
{
    const int i = blockIdx.x;
    if (i >= *count) return;
    const size_t q = (size_t)list[i], n = b.n, m = b.m, cap = b.cap;
    if (b.x) copy_span(b.x + q * n, r.x + (size_t)i * n, n);
    if (b.lam) copy_span(b.lam + q * m, r.lam + (size_t)i * m, m);
    const size_t mblk = (size_t)b.nblk * b.npair * 128;
    copy_span(b.Mblk + q * mblk, r.Mblk + (size_t)i * mblk, mblk);
    if (b.M32 && r.M32) { const size_t m32 = (size_t)b.nblk * b.nquad * 256; copy_span(b.M32 + q * m32, r.M32 + (size_t)i * m32, m32); }
    copy_span(b.Rinv + q * b.rtri, r.Rinv + (size_t)i * b.rtri, (size_t)b.rtri);
    copy_span(b.v + q * n, r.v + (size_t)i * n, n);
    copy_span(b.xunc + q * n, r.xunc + (size_t)i * n, n);
    copy_span(b.scaling + q * m, r.scaling + (size_t)i * m, m);
    copy_span(b.dupper + q * m, r.dupper + (size_t)i * m, m);
    copy_span(b.dlower + q * m, r.dlower + (size_t)i * m, m);
    copy_span(b.sense + q * m, r.sense + (size_t)i * m, m);
    copy_span(b.L + q * b.ltri, r.L + (size_t)i * b.ltri, (size_t)b.ltri);
    copy_span(b.vecs + q * 5 * cap, r.vecs + (size_t)i * 5 * cap, 5 * cap);
    copy_span(b.WS + q * cap, r.WS + (size_t)i * cap, cap);
    if (b.trace && r.trace && b.trace_cap == r.trace_cap) copy_span(b.trace + q * b.trace_cap, r.trace + (size_t)i * b.trace_cap, (size_t)b.trace_cap);
    if (threadIdx.x == 0) {
        b.qs[q] = r.qs[i];
        b.exitflag[q] = r.exitflag[i]; b.iter[q] = r.iter[i];
        if (b.fval) b.fval[q] = r.fval[i];
        if (b.soft) b.soft[q] = r.soft[i];
        if (b.fallback) b.fallback[q] = 0;
    }
}

} // namespace daqp_amd
