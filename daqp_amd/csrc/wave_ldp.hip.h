// wave_ldp.hip.h -- the dual active-set iteration of one LDP, executed by ONE wavefront.
//
// What it computes: the path daqp_ldp -> {compute_CSP, remove_blocking, compute_primal_and_fval,
// add_infeasible, update_LDL_add/remove, pivot_last, compute_singular_direction, refine_active,
// activate_constraints} of DAQP v0.9.1 (reference src/daqp.c:6-108, src/auxiliary.c,
// src/factorization.c).  How: designed for a 64-lane CDNA4 wavefront --
//   * lane <-> row of the working set for every triangular solve (column-oriented
//     substitution: one v_readlane broadcast + one LDS row read per step),
//   * lane <-> constraint row for the feasibility scan M*u (M streamed from HBM in a
//     row-blocked, k-pair-interleaved layout: one 16-byte load per lane, 1 KiB per wave),
//   * L (packed), D, the working set and a cache of the ACTIVE rows of M live in LDS,
//   * every floating-point accumulation keeps the reference's operation order (the file is
//     compiled with -ffp-contract=off), so a decision can only differ from the CPU reference
//     through the one documented tree reduction-free exception: none -- all sums are ordered.
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/daqp_amd.h"

namespace daqp_amd {
// A pointer read out of the descriptor (itself read through a pointer) is a GENERIC pointer to the compiler: every access through it is a
// flat_load / flat_store, which counts on the LDS counter as well, can be waited for only with "everything" (no partial vmcnt once a flat
// operation is pending) and orders itself against LDS copies in flight.  These are global memory:
#define DAQP_GLOBAL(T) __attribute__((address_space(1))) T
template <class T> __device__ __forceinline__ DAQP_GLOBAL(T) *as_global(T *p) { return (DAQP_GLOBAL(T) *)p; }
template <class T> __device__ __forceinline__ const DAQP_GLOBAL(T) *as_global(const T *p) { return (const DAQP_GLOBAL(T) *)p; }


constexpr int kEmpty = DAQP_EMPTY_IND;
constexpr int kBig = 0x7fffffff;

// One wavefront per workgroup: the LDS executes a wave's instructions in issue order, so a
// cross-lane write -> read hand-off needs no s_waitcnt/s_barrier, only that the COMPILER keeps
// the program order of the two LDS instructions (it reasons per thread and may otherwise swap a
// store to a[tid] with a load of a[tid+1]).  A wavefront-scope fence is exactly that.
#define WSYNC() do { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); } while (0)

__host__ __device__ constexpr __forceinline__ int tri(int k) { return (k * (k + 1)) >> 1; }
__device__ __forceinline__ int roff(int i, int n) { return ((2 * n - i - 1) * i) / 2; }
__device__ __forceinline__ int lane_id() { return (int)threadIdx.x; }

// Contiguous HBM -> LDS copy that bypasses the VGPRs: global_load_lds_dwordx4 (gfx950), 16 bytes per lane per
// instruction, every instruction of the copy in flight at once (one HBM round trip per tile instead of one
// per eight loads).  cnt even, src and dst 16-byte aligned; completion = vmcnt (copy_wait).
__device__ __forceinline__ void copy_async(double *dst, const double *src, int cnt)
{
    const int lane = lane_id(), pairs = cnt >> 1;
    for (int c = 0; c < pairs; c += 64)
        if (c + lane < pairs)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + 2 * (c + lane)),
                                             (__attribute__((address_space(3))) void *)(dst + 2 * c), 16, 0, 0);
}
// the same with 4 bytes per lane (256 B per instruction): for sources that are only 8-byte aligned
__device__ __forceinline__ void copy_async_dwords(double *dst, const double *src, int cnt)
{
    const int lane = lane_id(), nd = 2 * cnt;
    const unsigned *s32 = reinterpret_cast<const unsigned *>(src);
    unsigned *d32 = reinterpret_cast<unsigned *>(dst);
    for (int c = 0; c < nd; c += 64)
        if (c + lane < nd)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(s32 + c + lane),
                                             (__attribute__((address_space(3))) void *)(d32 + c), 4, 0, 0);
}
__device__ __forceinline__ void copy_wait() { __builtin_amdgcn_s_waitcnt(0x0F70); /* vmcnt(0) */ }

// dst[r*ld + c] = src[r*n + c] for r < rows, c < n: every lane walks the contiguous source with stride 64,
// keeps (row, col) incrementally (no division) and has 16 loads in flight before the first LDS store --
// at <= 3 waves per CU nothing else would hide the HBM latency of a load-store-load-store loop.
__device__ __forceinline__ void stage_rows(double *dst, const double *src_, int rows, int n, int ld)
{
    const DAQP_GLOBAL(double) *src = as_global(src_);      // (always global memory: the problem arrays)
    const int lane = lane_id(), total = rows * n;
    int r = 0, c = lane;
    while (c >= n) { c -= n; ++r; }
    constexpr int DEPTH = 16;
    for (int e0 = lane; e0 < total; e0 += 64 * DEPTH) {
        double v[DEPTH];
        int off[DEPTH];
#pragma unroll
        for (int q = 0; q < DEPTH; ++q) {
            const int e = e0 + 64 * q;
            v[q] = (e < total) ? src[e] : 0.0;
            off[q] = r * ld + c;
            c += 64;
            while (c >= n) { c -= n; ++r; }
        }
#pragma unroll
        for (int q = 0; q < DEPTH; ++q) if (e0 + 64 * q < total) dst[off[q]] = v[q];
    }
}

// broadcast lane `src` (wave-uniform) of v to every lane: two v_readlane_b32
__device__ __forceinline__ double rl(double v, int src)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_readlane(lo, src);
    hi = __builtin_amdgcn_readlane(hi, src);
    return __hiloint2double(hi, lo);
}
// element `idx` (wave-uniform) of a vector held as a[c] in lane (idx&63), chunk (idx>>6)
template <int C>
__device__ __forceinline__ double rlc(const double (&a)[C], int idx)
{
    double v = a[0];
#pragma unroll
    for (int c = 1; c < C; ++c)
        if ((idx >> 6) == c) v = a[c];
    return rl(v, idx & 63);
}

template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xF, 0xF, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double dmin2(double a, double b) { return b < a ? b : a; }

// wave-wide minimum of v (no NaNs expected): 4 DPP steps inside each row of 16, then 4 readlanes
__device__ __forceinline__ double wave_min(double v)
{
    v = dmin2(v, dpp_f64<0xB1>(v));   // quad_perm [1,0,3,2]
    v = dmin2(v, dpp_f64<0x4E>(v));   // quad_perm [2,3,0,1]
    v = dmin2(v, dpp_f64<0x141>(v));  // row_half_mirror
    v = dmin2(v, dpp_f64<0x140>(v));  // row_mirror
    double a = rl(v, 0), b = rl(v, 16), c = rl(v, 32), d = rl(v, 48);
    return dmin2(dmin2(a, b), dmin2(c, d));
}
// (value, index) argmin with lowest index on ties; lanes without a candidate pass idx = kBig.
// Returns the wave-uniform winner; idx stays kBig if nobody had one. `aux` rides along.
__device__ __forceinline__ void wave_argmin(double &v, int &idx, int &aux)
{
    const double vin = (idx == kBig) ? (double)DAQP_INF : v;
    const double mn = wave_min(vin);
    unsigned long long msk = __ballot(idx != kBig && vin == mn);
    int bi = kBig, ba = 0;
    while (msk) {                                   // one pass unless two lanes tie exactly
        const int l = __ffsll((long long)msk) - 1;
        const int ci = __builtin_amdgcn_readlane(idx, l);
        const int ca = __builtin_amdgcn_readlane(aux, l);
        if (ci < bi) { bi = ci; ba = ca; }
        msk &= msk - 1;
    }
    v = mn; idx = bi; aux = ba;
}

struct QState {   // per-problem scalars kept in HBM between launches
    int n_active, reuse_ind, sing_ind, iterations;
    int lam_swapped, setup_flag, need_activate, exitflag;
    double fval, soft_slack;
    int upd_flag, pad_;  // pad_: between k_setup and k_setup_m only -- what the first leaves to the second (setup_m.hip.h), else 0; upd_flag < 0: the last daqp_update_ldp(UPDATE_v|UPDATE_d) failed its bound check (utils.c:98-103): solves report it
                         // until the next update; factors and working set are kept (the reference's workspace stays usable too)
    int diag_h, n_prox; // n_prox > 0: the factor is of a shifted Hessian, solves go through the proximal outer loop (prox.hip.h)
                        // diag_h 1: H was diagonal -- the reference's RinvD branch (utils.c:245-312): rows < ms of R^-1 are kept
                        // un-normalised (their image in M is the exact unit vector) and x is not divided by the scaling
};

// One wave's view of one LDP.  L and rowc are LDS (or HBM scratch when spilled); the small
// vectors are always LDS; M and the bounds are read-only HBM.
template <int C, int NB, int NP>
struct Wave {
    // sizes
    int n, m, ms, cap, npair, nblk, ldr;
    // LDS / scratch
    double *L, *rowc;
    double *D, *xl, *zl, *lam, *lams, *u;
    int *ws, *sense, *pend_id;
    double *pend_lam;
    // HBM, read-only during the iteration
    const double *Mblk, *dupper, *dlower, *scaling;
    // uniform iterate state
    int na, reuse, sing, has_soft;
    double fval, soft;
    DAQPSettings st;
    // optional event trace (+id+1 add, -(id+1) remove)
    int *trace; int trace_cap, trace_len;
    // the whole constraint matrix of this QP, register-resident (NB row blocks x NP k-pairs per
    // lane; NB == 0: streamed from HBM instead).  Padding pairs/rows are zero.
    // optional phase cycle counters (s_memtime): csp, blocking, primal, scan, add, remove, other
    long long prof[8];
    bool profiling;
    // settings->time_limit > 0 (daqp.c:95-103): ticks of the constant device clock at the start of this problem's daqp_solve
    // (solve_stamp: one stamp per problem and daqp_batch_solve, shared by every launch of that solve) and seconds per tick
    unsigned long long t_start;
    double tick_s;
};

// s_memrealtime ticks since t_start against the limit; tick_s = seconds per tick (hipDeviceAttributeWallClockRate, queried by
// the host at batch creation: 100 MHz on gfx950)
__device__ __forceinline__ bool time_is_up(unsigned long long t_start, double limit_s, double tick_s)
{
    const unsigned long long now = __builtin_amdgcn_s_memrealtime();
    return (double)(now - t_start) * tick_s > limit_s;
}
// The reference starts ONE timer per daqp_solve and shares it across everything that solve does -- the proximal outer loop
// included (daqp.c:95-103, api.c:8-59).  Here a solve may take several launches (workgroup kernel -> one-wave fallback; two
// launches per outer iteration of the proximal loop): the first launch that touches problem q in a daqp_batch_solve stamps
// tstart[q] (zeroed by the host at the start of the call when the limit is armed), the later ones inherit it.
__device__ __forceinline__ unsigned long long solve_stamp(unsigned long long *tstart, int q)
{
    const unsigned long long now = __builtin_amdgcn_s_memrealtime();
    if (tstart == nullptr) return now;
    const unsigned long long t = __hip_atomic_load(tstart + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (t != 0) return t;
    if (lane_id() == 0) __hip_atomic_store(tstart + q, now, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return now;
}

#define PROF_T0(w) long long prof_t0_ = (w).profiling ? (long long)__builtin_readcyclecounter() : 0
#define PROF_ACC(w, slot) do { if ((w).profiling) { const long long t1_ = (long long)__builtin_readcyclecounter(); (w).prof[slot] += t1_ - prof_t0_; prof_t0_ = t1_; } } while (0)

// branch markers in the event trace (tests count them; daqp_batch_read_trace hands them out with the adds / removes):
// pivot_last swapped a constraint out (auxiliary.c:379-396), a singular direction was followed (auxiliary.c:357-376),
// the active rows were refined (auxiliary.c:498-593), the factor was rebuilt at a KKT point (daqp.c:33-46) or by the
// cycle guard (daqp.c:66-85)
enum : int { kTraceMark = 0x40000000, kTracePivot = kTraceMark + 1, kTraceSingular = kTraceMark + 2, kTraceRefine = kTraceMark + 3,
             kTraceRefactor = kTraceMark + 4, kTraceCycleReset = kTraceMark + 5 };

template <int C, int NB, int NP>
__device__ __forceinline__ void trace_ev(Wave<C, NB, NP> &w, int ev)
{
    if (w.trace) {
        if (lane_id() == 0 && w.trace_len < w.trace_cap) w.trace[w.trace_len] = ev;
        w.trace_len++;
    }
}

// rowc[slot][0..n) <- row `id` of the LDP constraint matrix, gathered from the blocked HBM layout
// [row/64][k/2][row%64][k%2] (the register-resident variant is k_ldp_reg / wave_ldp_reg.hip.h).
// Simple-bound rows are stored densely with a zero prefix.
template <int C, int NB, int NP>
__device__ __forceinline__ void fetch_row(Wave<C, NB, NP> &w, int id, int slot)
{
    const int lane = lane_id();
    double *dst = w.rowc + (size_t)slot * w.ldr;
    const double2 *src = reinterpret_cast<const double2 *>(w.Mblk) + ((size_t)(id >> 6) * w.npair) * 64 + (id & 63);
    for (int t = lane; t < w.npair; t += 64) {
        const double2 v = src[(size_t)t * 64];
        dst[2 * t] = v.x;
        if (2 * t + 1 < w.n) dst[2 * t + 1] = v.y;
    }
    WSYNC();
}

constexpr int kPre = 8;   // columns / rows fetched ahead of each dependency chain in this (LDS- or HBM-resident) variant

// factorization.c:4-15 (4 interleaved partial sums, then (s0+s1)+(s2+s3))
__device__ __forceinline__ double dot4(const double *a, const double *b, int len)
{
    double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
    int i = 0;
    for (; i + 15 < len; i += 16) {   // 32 loads in flight before the four chains consume them: when the rows live in HBM
        double x[16], y[16];          // (L / row cache spilled, n = 200) a trip per four elements is pure latency
#pragma unroll
        for (int q = 0; q < 16; ++q) { x[q] = a[i + q]; y[q] = b[i + q]; }
#pragma unroll
        for (int q = 0; q < 16; q += 4) {
            s0 += x[q] * y[q]; s1 += x[q + 1] * y[q + 1]; s2 += x[q + 2] * y[q + 2]; s3 += x[q + 3] * y[q + 3];
        }
    }
    for (; i + 3 < len; i += 4) {
        s0 += a[i] * b[i];
        s1 += a[i + 1] * b[i + 1];
        s2 += a[i + 2] * b[i + 2];
        s3 += a[i + 3] * b[i + 3];
    }
    for (; i < len; i++) s0 += a[i] * b[i];
    return (s0 + s1) + (s2 + s3);
}

// ---------------------------------------------------------------------------------------
// LDL' row append (factorization.c:21-111): Gram column by lane<->active row, forward
// substitution column by column, ordered Schur complement
// ---------------------------------------------------------------------------------------
template <int C, int NB, int NP>
__device__ __forceinline__ void ldl_append(Wave<C, NB, NP> &w, int id)
{
    const int lane = lane_id(), na = w.na, n = w.n, base = tri(na);
    fetch_row(w, id, na);
    const int c0 = id < w.ms ? id : 0;
    w.sing = kEmpty;
    const double *Mi = w.rowc + (size_t)na * w.ldr;
    double g[C];
    int ns_act = 0;
#pragma unroll
    for (int c = 0; c < C; ++c) {
        const int k = lane + 64 * c;
        g[c] = 0;
        int soft_k = 0;
        if (k <= na) {
            const int idk = (k < na) ? w.ws[k] : id;
            const int j = (k < na && idk < w.ms) ? (c0 > idk ? c0 : idk) : c0;
            g[c] = dot4(w.rowc + (size_t)k * w.ldr + j, Mi + j, n - j);
            soft_k = (w.sense[idk] & DAQP_SOFT) ? 1 : 0;
        }
        if (w.has_soft) ns_act += __popcll(__ballot(soft_k));
    }
    double dnew = rlc<C>(g, na);
    if (w.sense[id] & DAQP_SOFT) dnew += w.st.rho_soft;
    if (na == 0) {
        if (lane == 0) w.D[0] = dnew;
        WSYNC();
        return;
    }
    // forward substitution  l <- L \ g   (factorization.c:81-88)
    for (int j0 = 0; j0 < na - 1; j0 += kPre) {   // the lane's own L entries of kPre columns are loaded before their chain
        double Lv[kPre][C];
#pragma unroll
        for (int q = 0; q < kPre; ++q)
#pragma unroll
            for (int c = 0; c < C; ++c) {
                const int k = lane + 64 * c, j = j0 + q;
                Lv[q][c] = w.L[(k > j && k < na) ? tri(k) + j : 0];   // clamped address, unconditional load: no branch per load
            }
#pragma unroll
        for (int q = 0; q < kPre; ++q) {
            const int j = j0 + q;
            if (j < na - 1) {
                const double lj = rlc<C>(g, j);
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    const int k = lane + 64 * c;
                    if (k > j && k < na) g[c] -= Lv[q][c] * lj;
                }
            }
        }
    }
    // l_k /= D_k ; d_new -= sum_k l_k^2 D_k, in k order (factorization.c:93-103)
    double p[C];
#pragma unroll
    for (int c = 0; c < C; ++c) {
        const int k = lane + 64 * c;
        p[c] = 0;
        if (k < na) {
            const double t = g[c];
            const double lk = t / w.D[k];
            w.L[base + k] = lk;
            p[c] = t * lk;
        }
    }
    double acc = dnew;
    for (int k = 0; k < na; ++k) acc -= rlc<C>(p, k);
    int sing = kEmpty;
    if (acc < w.st.sing_tol || na >= n + ns_act) { sing = na; acc = 0; }
    if (lane == 0) w.D[na] = acc;
    w.sing = sing;
    WSYNC();
}

// ---------------------------------------------------------------------------------------
// LDL' row delete (factorization.c:112-151): staged compaction of packed L, then the
// Gill-Golub-Murray-Saunders C1 rank-one update with lane<->trailing row
// ---------------------------------------------------------------------------------------
template <int C, int NB, int NP>
__device__ __forceinline__ void ldl_delete(Wave<C, NB, NP> &w, int r)
{
    const int lane = lane_id(), na = w.na;
    if (na == r + 1) return;
    const int nupd = na - r - 1;
    double wv[C];
#pragma unroll
    for (int c = 0; c < C; ++c) {
        const int t = lane + 64 * c;
        wv[c] = (t < nupd) ? w.L[tri(r + 1 + t) + r] : 0.0;
    }
    // move rows r+1.. up by one and drop column r.  Destination e always reads from a higher
    // address, so ascending chunks with "read all, then write all" never clobber a live source.
    const int e0 = tri(r), e1 = tri(na - 1);
    constexpr int U = 4;
    for (int cb = e0; cb < e1; cb += 64 * U) {
        double tmp[U];
#pragma unroll
        for (int q = 0; q < U; ++q) {
            const int e = cb + q * 64 + lane;
            tmp[q] = 0;
            if (e < e1) {
                int i = (int)((sqrtf(8.0f * (float)e + 1.0f) - 1.0f) * 0.5f);
                while (tri(i + 1) <= e) ++i;
                while (tri(i) > e) --i;
                const int j = e - tri(i);
                tmp[q] = w.L[tri(i + 1) + j + (j >= r ? 1 : 0)];
            }
        }
        WSYNC();
#pragma unroll
        for (int q = 0; q < U; ++q) {
            const int e = cb + q * 64 + lane;
            if (e < e1) w.L[e] = tmp[q];
        }
        WSYNC();
    }
    double alpha = w.D[r];
    for (int j = 0; j < nupd; ++j) {
        const int i = r + 1 + j;
        const double p = rlc<C>(wv, j);
        const double Di = w.D[i];
        const double dbar = Di + alpha * p * p;
        const double beta = p * alpha / dbar;
        alpha = Di * alpha / dbar;
        if (lane == 0) w.D[i - 1] = dbar;
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const int t = lane + 64 * c;
            if (t > j && t < nupd) {
                const int pos = tri(r + t) + r + j;
                const double l = w.L[pos];
                wv[c] -= p * l;
                w.L[pos] = l + beta * wv[c];
            }
        }
    }
    WSYNC();
}

// auxiliary.c:3-22 without the trailing pivot; returns 1 if the factor became singular
template <int C, int NB, int NP>
__device__ __forceinline__ int drop_core(Wave<C, NB, NP> &w, int r)
{
    const int lane = lane_id();
    const int idr = w.ws[r];
    trace_ev(w, -(idr + 1));
    if (lane == 0) w.sense[idr] &= ~DAQP_ACTIVE;
    ldl_delete(w, r);
    w.na--;
    int wsn[C];
    double lmn[C];
#pragma unroll
    for (int c = 0; c < C; ++c) {
        const int i = lane + 64 * c;
        wsn[c] = 0; lmn[c] = 0;
        if (i >= r && i < w.na) { wsn[c] = w.ws[i + 1]; lmn[c] = w.lam[i + 1]; }
    }
    WSYNC();
#pragma unroll
    for (int c = 0; c < C; ++c) {
        const int i = lane + 64 * c;
        if (i >= r && i < w.na) { w.ws[i] = wsn[c]; w.lam[i] = lmn[c]; }
    }
    // the active-row cache is indexed by working-set position: close the gap (each lane moves
    // its own columns, rows ascending, so no lane reads what another lane wrote)
    for (int i = r; i < w.na; ++i) {
        const double *src = w.rowc + (size_t)(i + 1) * w.ldr;
        double *dst = w.rowc + (size_t)i * w.ldr;
        for (int j = lane; j < w.n; j += 64) dst[j] = src[j];
    }
    if (r < w.reuse) w.reuse = r;
    int took = 0;
    if (w.na > 0 && w.D[w.na - 1] < w.st.sing_tol) {
        w.sing = w.na - 1;
        took = 1;
    }
    WSYNC();
    if (took && lane == 0) w.D[w.na - 1] = 0;
    WSYNC();
    return took;
}

template <int C, int NB, int NP>
__device__ __forceinline__ void push_core(Wave<C, NB, NP> &w, int id, double lamv) // auxiliary.c:27-40
{
    const int lane = lane_id();
    trace_ev(w, id + 1);
    if (lane == 0) w.sense[id] |= DAQP_ACTIVE;
    WSYNC();
    ldl_append(w, id);
    if (lane == 0) { w.ws[w.na] = id; w.lam[w.na] = lamv; }
    w.na++;
    WSYNC();
}

// daqp_pivot_last (auxiliary.c:379-396): its recursion through remove/add_constraint becomes
// an explicit stack (in LDS) of constraints waiting to be re-inserted
template <int C, int NB, int NP>
__device__ __forceinline__ void pivot_tail(Wave<C, NB, NP> &w)
{
    const int lane = lane_id();
    int depth = 0;
    for (;;) {
        const int r = w.na - 2;
        bool piv = false;
        if (w.na > 1) {
            const double dr = w.D[r], dl = w.D[w.na - 1];
            piv = dr < w.st.pivot_tol && dr < dl;
        }
        if (piv) {
            trace_ev(w, kTracePivot);
            if (lane == 0) { w.pend_id[depth] = w.ws[r]; w.pend_lam[depth] = w.lam[r]; }
            depth++;
            WSYNC();
            if (drop_core(w, r)) break;
            continue;
        }
        if (depth == 0) break;
        if (w.sing != kEmpty) break;
        depth--;
        const int id = w.pend_id[depth];
        const double lv = w.pend_lam[depth];
        push_core(w, id, lv);
    }
}

template <int C, int NB, int NP>
__device__ __forceinline__ void remove_constraint(Wave<C, NB, NP> &w, int r)
{
    if (!drop_core(w, r)) pivot_tail(w);
}
template <int C, int NB, int NP>
__device__ __forceinline__ void add_constraint(Wave<C, NB, NP> &w, int id, double lamv)
{
    push_core(w, id, lamv);
    pivot_tail(w);
}

// ---------------------------------------------------------------------------------------
// constrained stationary point: L D L' lam* = -d_k  (auxiliary.c:314-354)
// ---------------------------------------------------------------------------------------
template <int C, int NB, int NP>
__device__ __forceinline__ void forward_rows(Wave<C, NB, NP> &w, double (&acc)[C], int from)
{
    // acc[c] holds the right-hand side of rows >= from; rows < from are final in xl
    const int lane = lane_id(), na = w.na;
    for (int j0 = 0; j0 < na - 1; j0 += kPre) {   // kPre columns of L per trip (loads first, then the ordered chain)
        double Lv[kPre][C];
#pragma unroll
        for (int q = 0; q < kPre; ++q)
#pragma unroll
            for (int c = 0; c < C; ++c) {
                const int i = lane + 64 * c, j = j0 + q;
                Lv[q][c] = w.L[(i >= from && i > j && i < na) ? tri(i) + j : 0];
            }
#pragma unroll
        for (int q = 0; q < kPre; ++q) {
            const int j = j0 + q;
            if (j < na - 1) {
                const double xj = (j < from) ? w.xl[j] : rlc<C>(acc, j);
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    const int i = lane + 64 * c;
                    if (i >= from && i > j && i < na) acc[c] -= Lv[q][c] * xj;
                }
            }
        }
    }
}
// b <- L' \ b for the leading `cnt` rows; multiplication order as the reference: b_j * L[j][i]
template <int C, int NB, int NP>
__device__ __forceinline__ void backward_rows(Wave<C, NB, NP> &w, double (&b)[C], int cnt)
{
    const int lane = lane_id();
    for (int j0 = cnt - 1; j0 >= 1; j0 -= kPre) {
        double Lv[kPre][C];
#pragma unroll
        for (int q = 0; q < kPre; ++q)
#pragma unroll
            for (int c = 0; c < C; ++c) {
                const int i = lane + 64 * c, j = j0 - q;
                Lv[q][c] = w.L[(j >= 1 && i < j) ? tri(j) + i : 0];
            }
#pragma unroll
        for (int q = 0; q < kPre; ++q) {
            const int j = j0 - q;
            if (j >= 1) {
                const double bj = rlc<C>(b, j);
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    const int i = lane + 64 * c;
                    if (i < j) b[c] -= bj * Lv[q][c];
                }
            }
        }
    }
}

template <int C, int NB, int NP>
__device__ __forceinline__ void solve_csp(Wave<C, NB, NP> &w)
{
    const int lane = lane_id(), na = w.na, from = w.reuse;
    double acc[C];
#pragma unroll
    for (int c = 0; c < C; ++c) {
        const int i = lane + 64 * c;
        acc[c] = 0;
        if (i >= from && i < na) {
            const int id = w.ws[i];
            acc[c] = (w.sense[id] & DAQP_LOWER) ? -w.dlower[id] : -w.dupper[id];
        }
    }
    forward_rows<C>(w, acc, from);
    double b[C];
#pragma unroll
    for (int c = 0; c < C; ++c) {
        const int i = lane + 64 * c;
        b[c] = 0;
        if (i < na) {
            if (i >= from) {
                w.xl[i] = acc[c];
                b[c] = acc[c] / w.D[i];
                w.zl[i] = b[c];
            } else b[c] = w.zl[i];
        }
    }
    backward_rows<C>(w, b, na);
#pragma unroll
    for (int c = 0; c < C; ++c) {
        const int i = lane + 64 * c;
        if (i < na) w.lams[i] = b[c];
    }
    w.reuse = na;
    WSYNC();
}

template <int C, int NB, int NP>
__device__ __forceinline__ void singular_direction(Wave<C, NB, NP> &w) // auxiliary.c:357-376
{
    const int lane = lane_id(), s = w.sing, base = tri(s);
    double b[C];
#pragma unroll
    for (int c = 0; c < C; ++c) {
        const int i = lane + 64 * c;
        b[c] = (i < s) ? -w.L[base + i] : 0.0;
    }
    backward_rows<C>(w, b, s);
    const bool flip = (w.sense[w.ws[s]] & DAQP_LOWER) != 0;
#pragma unroll
    for (int c = 0; c < C; ++c) {
        const int i = lane + 64 * c;
        if (i <= s) {
            double v = (i == s) ? 1.0 : b[c];
            w.lams[i] = flip ? -v : v;
        }
    }
    WSYNC();
}

// auxiliary.c:277-311 (SOFT_WEIGHTS off): ratio test over the working set, drop the argmin
template <int C, int NB, int NP>
__device__ __forceinline__ int remove_blocking(Wave<C, NB, NP> &w)
{
    const int lane = lane_id(), na = w.na;
    const double dtol = w.st.dual_tol;
    const bool regular = (w.sing == kEmpty);
    double bv = DAQP_INF;
    int bi = kBig, aux = 0;
    double lm[C], ls[C];
#pragma unroll
    for (int c = 0; c < C; ++c) {
        const int i = lane + 64 * c;
        lm[c] = 0; ls[c] = 0;
        if (i < na) {
            lm[c] = w.lam[i]; ls[c] = w.lams[i];
            const int sn = w.sense[w.ws[i]];
            bool blocking = !(sn & DAQP_IMMUTABLE);
            if (sn & DAQP_LOWER) { if (ls[c] < dtol) blocking = false; }
            else if (ls[c] > -dtol) blocking = false;
            if (blocking) {
                const double cand = regular ? -lm[c] / (ls[c] - lm[c]) : -lm[c] / ls[c];
                if (cand < bv) { bv = cand; bi = i; }
            }
        }
    }
    wave_argmin(bv, bi, aux);
    if (bi == kBig) return 0;
    const double alpha = bv;
#pragma unroll
    for (int c = 0; c < C; ++c) {
        const int i = lane + 64 * c;
        if (i < na) w.lam[i] = regular ? lm[c] + alpha * (ls[c] - lm[c]) : lm[c] + alpha * ls[c];
    }
    w.sing = kEmpty;
    WSYNC();
    remove_constraint(w, bi);
    return 1;
}

// u = -M_k' lam*, fval = rho*sum_soft lam*^2 + |u|^2 (auxiliary.c:46-88); every sum is in
// the reference's order: over the working set for u, over j for |u|^2
template <int C, int NB, int NP>
__device__ __forceinline__ void primal_u(Wave<C, NB, NP> &w)
{
    const int lane = lane_id(), na = w.na, n = w.n;
    double uu[C];
#pragma unroll
    for (int c = 0; c < C; ++c) uu[c] = 0;
    for (int i0 = 0; i0 < na; i0 += kPre) {   // kPre cached rows per trip
        double rv[kPre][C], li[kPre];
#pragma unroll
        for (int q = 0; q < kPre; ++q) {
            const int i = i0 + q;
            li[q] = (i < na) ? w.lams[i] : 0.0;
            const double *row = w.rowc + (size_t)(i < na ? i : 0) * w.ldr;
#pragma unroll
            for (int c = 0; c < C; ++c) {
                const int j = lane + 64 * c;
                rv[q][c] = row[(j < n) ? j : 0];
            }
        }
#pragma unroll
        for (int q = 0; q < kPre; ++q)
            if (i0 + q < na) {
#pragma unroll
                for (int c = 0; c < C; ++c) uu[c] -= rv[q][c] * li[q];
            }
    }
#pragma unroll
    for (int c = 0; c < C; ++c) {
        const int j = lane + 64 * c;
        if (j < n) w.u[j] = uu[c];
    }
    double fv = 0;
    if (w.has_soft) {
        for (int i = 0; i < na; ++i)
            if (w.sense[w.ws[i]] & DAQP_SOFT) { const double li = w.lams[i]; fv += li * li; }
    }
    fv = fv * w.st.rho_soft;
    w.soft = fv;
    WSYNC();
}
template <int C, int NB, int NP>
__device__ __forceinline__ double ordered_norm2(Wave<C, NB, NP> &w, double start)
{
    double fv = start;
    for (int j = 0; j < w.n; ++j) { const double uj = w.u[j]; fv += uj * uj; }
    return fv;
}

// feasibility scan + most-violated pick (auxiliary.c:89-198).  Returns the row (or kBig) and
// sets `upper`.  When `with_fval`, |u|^2 (j-ordered) is accumulated alongside.
template <int C, int NB, int NP>
__device__ __forceinline__ int scan_rows(Wave<C, NB, NP> &w, int &upper, bool with_fval)
{
    const int lane = lane_id(), n = w.n;
    const double ep = -w.st.primal_tol;
    double bv = 0.0;
    int bi = kBig, bup = 0;
    double fv = w.soft;
    const double2 *u2 = reinterpret_cast<const double2 *>(w.u);
    const bool odd = (n & 1) != 0;
    for (int blk = 0; blk < w.nblk; ++blk) {
        const int r = blk * 64 + lane;
        const double2 *src = reinterpret_cast<const double2 *>(w.Mblk) + ((size_t)blk * w.npair) * 64 + lane;
        double mu = 0;
        const int full = odd ? w.npair - 1 : w.npair;
        if (r < w.m) {
            // the stream of M: 16 x 16-byte loads per lane in flight (16 KiB per wave) before the k-ordered chain
            // consumes them -- with a handful in flight the scan is pure HBM latency at large n
            int t = 0;
            for (; t + 16 <= full; t += 16) {
                double2 mm[16], uk[16];
#pragma unroll
                for (int q = 0; q < 16; ++q) mm[q] = src[(size_t)(t + q) * 64];
#pragma unroll
                for (int q = 0; q < 16; ++q) uk[q] = u2[t + q];
#pragma unroll
                for (int q = 0; q < 16; ++q) { mu += mm[q].x * uk[q].x; mu += mm[q].y * uk[q].y; }
            }
            if (t < full) {
                double2 mm[16], uk[16];
#pragma unroll
                for (int q = 0; q < 16; ++q) { const int tt = (t + q < full) ? t + q : full - 1; mm[q] = src[(size_t)tt * 64]; uk[q] = u2[tt]; }
#pragma unroll
                for (int q = 0; q < 16; ++q) if (t + q < full) { mu += mm[q].x * uk[q].x; mu += mm[q].y * uk[q].y; }
            }
            if (odd) mu += src[(size_t)full * 64].x * w.u[n - 1];
        }
        if (with_fval && blk == 0) {
            for (int j = 0; j < n; ++j) { const double uj = w.u[j]; fv += uj * uj; }
        }
        if (r < w.m) {
            const int sn = w.sense[r];
            if (!(sn & (DAQP_ACTIVE + DAQP_IMMUTABLE))) {
                const double bound = ep * w.scaling[r];
                double cand = w.dupper[r] - mu;
                if (cand < bv && cand < bound) { bv = cand; bi = r; bup = 1; }
                else {
                    cand = mu - w.dlower[r];
                    if (cand < bv && cand < bound) { bv = cand; bi = r; bup = 0; }
                }
            }
        }
    }
    if (with_fval && w.nblk == 0) fv = ordered_norm2(w, fv);

    if (with_fval) w.fval = fv;
    wave_argmin(bv, bi, bup);
    upper = bup;
    return bi;
}

template <int C, int NB, int NP>
__device__ __forceinline__ void commit_add(Wave<C, NB, NP> &w, int pick, int upper) // auxiliary.c:152-166
{
    if (lane_id() == 0) {
        if (upper) w.sense[pick] &= ~DAQP_LOWER; else w.sense[pick] |= DAQP_LOWER;
    }
    double *t = w.lam; w.lam = w.lams; w.lams = t;
    WSYNC();
    add_constraint(w, pick, upper ? 1.0 : -1.0);
}

// one step of iterative refinement on the active rows (auxiliary.c:498-593)
template <int C, int NB, int NP>
__device__ __forceinline__ void refine_active(Wave<C, NB, NP> &w)
{
    const int lane = lane_id(), na = w.na, n = w.n;
    w.reuse = 0;
    double acc[C];
#pragma unroll
    for (int c = 0; c < C; ++c) {
        const int i = lane + 64 * c;
        acc[c] = 0;
        if (i < na) {
            const int id = w.ws[i];
            const double *row = w.rowc + (size_t)i * w.ldr;
            double mu = 0;
            for (int j = (id < w.ms ? id : 0); j < n; ++j) mu += row[j] * w.u[j];
            const double d = (w.sense[id] & DAQP_LOWER) ? w.dlower[id] : w.dupper[id];
            acc[c] = mu - d;
            if (w.sense[id] & DAQP_SOFT) acc[c] -= w.st.rho_soft * w.lams[i];
        }
    }
    forward_rows<C>(w, acc, 0);
    double b[C];
#pragma unroll
    for (int c = 0; c < C; ++c) {
        const int i = lane + 64 * c;
        b[c] = 0;
        if (i < na) { w.xl[i] = acc[c]; b[c] = acc[c] / w.D[i]; w.zl[i] = b[c]; }
    }
    backward_rows<C>(w, b, na);
#pragma unroll
    for (int c = 0; c < C; ++c) {
        const int i = lane + 64 * c;
        if (i < na) { w.xl[i] = b[c]; w.lams[i] += b[c]; }
    }
    WSYNC();
    double uu[C];
#pragma unroll
    for (int c = 0; c < C; ++c) { const int j = lane + 64 * c; uu[c] = (j < n) ? w.u[j] : 0.0; }
    for (int i = 0; i < na; ++i) {
        const double dl = w.xl[i];
        const int id = w.ws[i];
        const int j0 = id < w.ms ? id : 0;
        const double *row = w.rowc + (size_t)i * w.ldr;
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const int j = lane + 64 * c;
            if (j < n && j >= j0) uu[c] -= row[j] * dl;
        }
    }
    WSYNC();
#pragma unroll
    for (int c = 0; c < C; ++c) { const int j = lane + 64 * c; if (j < n) w.u[j] = uu[c]; }
    WSYNC();
    w.fval = ordered_norm2(w, w.soft);
}

// (re)build the working set from the ACTIVE bits, in index order (auxiliary.c:399-479)
template <int C, int NB, int NP>
__device__ __forceinline__ int activate_marked(Wave<C, NB, NP> &w)
{
    const int lane = lane_id();
    for (int blk = 0; blk * 64 < w.m; ++blk) {
        const int r = blk * 64 + lane;
        unsigned long long msk = __ballot(r < w.m && (w.sense[r] & DAQP_ACTIVE));
        while (msk) {
            const int i = blk * 64 + __ffsll((long long)msk) - 1;
            msk &= msk - 1;
            add_constraint(w, i, (w.sense[i] & DAQP_LOWER) ? -1.0 : 1.0);
            if (w.sing == kEmpty) continue;
            const int last = w.ws[w.na - 1];
            if (w.sense[last] & DAQP_IMMUTABLE) {
                // a new equality depends on the active ones: consistent => ignore it
                singular_direction(w);
                double resid = 0.0, scale = 1.0;
                for (int j = 0; j < w.na; ++j) {
                    const int id = w.ws[j];
                    const double bd = (w.sense[id] & DAQP_LOWER) ? w.dlower[id] : w.dupper[id];
                    const double t = w.lams[j] * bd;
                    resid += t;
                    scale += t < 0 ? -t : t;
                }
                WSYNC();
                if (lane == 0) w.sense[last] &= ~DAQP_ACTIVE;
                w.na--;
                w.sing = kEmpty;
                if (w.reuse > w.na) w.reuse = w.na;
                WSYNC();
                if (resid <= w.st.primal_tol * scale && resid >= -w.st.primal_tol * scale) continue;
                return DAQP_EXIT_OVERDETERMINED_INITIAL;
            }
            int flag = 1;
            for (int q = i; q < w.m; q += 1) {
                // rows after i: equalities that could not be activated are an error, the rest are cleaned
                const int sn = w.sense[q];
                if (sn & DAQP_ACTIVE) {
                    if (sn & DAQP_IMMUTABLE) flag = DAQP_EXIT_OVERDETERMINED_INITIAL;
                    else if (lane == 0) w.sense[q] = sn & ~DAQP_ACTIVE;
                }
            }
            w.na--;
            w.sing = kEmpty;
            WSYNC();
            return flag;
        }
    }
    return 1;
}

template <int C, int NB, int NP>
__device__ __forceinline__ void reset_ws(Wave<C, NB, NP> &w) { w.sing = kEmpty; w.na = 0; w.reuse = 0; }

// ---------------------------------------------------------------------------------------
// daqp_ldp (daqp.c:6-108)
// ---------------------------------------------------------------------------------------
template <int C, int NB, int NP>
__device__ __forceinline__ int ldp_loop(Wave<C, NB, NP> &w, int &iterations)
{
    const int lane = lane_id();
    int flag = DAQP_EXIT_ITERLIMIT, it, repaired = 0, stall = 0;
    double best = -1;
    const double fbound = 2 * w.st.fval_bound;
    for (it = 1; it < w.st.iter_limit; ++it) {
        PROF_T0(w);
        if (w.sing == kEmpty) {
            solve_csp(w);
            PROF_ACC(w, 0);
            const int blocked = remove_blocking(w);
            if (blocked) PROF_ACC(w, 5); else PROF_ACC(w, 1);
            if (blocked) {   // falls through to the end of the reference's loop body: the clock check applies
                if (w.st.time_limit > 0 && (it & 31) == 0 && time_is_up(w.t_start, w.st.time_limit, w.tick_s)) { flag = DAQP_EXIT_TIMELIMIT; break; }
                continue;
            }
            primal_u(w);
            PROF_ACC(w, 2);
            int upper = 0;
            int pick = scan_rows(w, upper, true);
            PROF_ACC(w, 3);
            if (w.fval > fbound) { flag = DAQP_EXIT_INFEASIBLE; break; }
            if (pick == kBig) {
                double dmin = w.D[0];
                for (int i = 1; i < w.na; ++i) { const double di = w.D[i]; if (di < dmin) dmin = di; }
                if (w.na > 2 && repaired != 1 && dmin < w.st.refactor_tol) {
                    repaired = 1;
                    trace_ev(w, kTraceRefactor);
                    for (int i = lane; i < w.na; i += 64) {
                        const int id = w.ws[i];
                        if (w.lam[i] >= 0) w.sense[id] &= ~DAQP_LOWER; else w.sense[id] |= DAQP_LOWER;
                    }
                    WSYNC();
                    reset_ws(w);
                    activate_marked(w);
                    continue;
                }
                if (w.na > 0 && dmin < w.st.pivot_tol) {
                    trace_ev(w, kTraceRefine);
                    refine_active(w);
                    pick = scan_rows(w, upper, false);
                    if (pick != kBig) { commit_add(w, pick, upper); continue; }
                }
                flag = (w.soft > w.st.primal_tol) ? DAQP_EXIT_SOFT_OPTIMAL : DAQP_EXIT_OPTIMAL;
                break;
            }
            commit_add(w, pick, upper);
            PROF_ACC(w, 4);
            if (w.fval - best < w.st.progress_tol) {
                if (stall++ > w.st.cycle_tol) {
                    if (repaired == 1) { flag = DAQP_EXIT_CYCLE; break; }
                    repaired = 1;
                    trace_ev(w, kTraceCycleReset);
                    reset_ws(w);
                    activate_marked(w);
                    stall = 0;
                    best = -1;
                }
            } else { best = w.fval; stall = 0; }
        } else {
            trace_ev(w, kTraceSingular);
            singular_direction(w);
            if (!remove_blocking(w)) { flag = DAQP_EXIT_INFEASIBLE; break; }
        }
        // daqp.c:95-103: every 32nd iteration that reaches the end of the loop body looks at the clock
        if (w.st.time_limit > 0 && (it & 31) == 0 && time_is_up(w.t_start, w.st.time_limit, w.tick_s)) { flag = DAQP_EXIT_TIMELIMIT; break; }
    }
    iterations = it;
    return flag;
}

} // namespace daqp_amd
