// wg_ldp.hip.h -- the dual active-set iteration for LARGE problems (working sets beyond 64 rows, n up to 255): one
// WORKGROUP of W wavefronts per QP, persistent workgroups pulling problems from an atomic counter.
//
// Same algorithm as wave_ldp.hip.h (reference src/daqp.c:6-108, src/auxiliary.c, src/factorization.c) and, in exact mode,
// the same floating-point operation order.  What the round-1 profile of config C4 (n=200, m=600) said: the one-wave kernel
// with L and the active rows in HBM scratch moved ~1.3 MB per iteration (M 960 KB + two passes over the active-row cache +
// two over L) at latency-bound rates.  Here
//   * packed L lives in LDS (up to capL rows; a problem whose working set outgrows that is handed to the one-wave kernel),
//     so every triangular solve / rank-one update of the iteration is LDS + cross-lane traffic on wave 0 (the "master");
//   * the bandwidth phases are spread over all W waves: the feasibility scan M u (lane <-> constraint row, every wave its
//     own 64-row blocks, 16 KB of loads in flight per wave), the primal step u = -M_W' lam* (lane <-> column) and the Gram
//     column M_W m_new of the LDL' append (lane <-> active row, read from a TRANSPOSED copy of the active rows so that the
//     loads coalesce), plus the element-parallel compaction of L on a removal;
//   * the active rows are kept in both orientations in a per-WORKGROUP scratch (not per problem): a few hundred KB that
//     stay cache-resident because only as many problems are in flight as there are resident workgroups.
// The master runs the reference's control flow unchanged and hands the parallel phases to the other waves through a
// command word in LDS (post, s_barrier, everybody works, s_barrier).
#pragma once
#include "wave_ldp.hip.h"

namespace daqp_amd {

enum : int { WG_EXIT = 0, WG_PRIMAL = 1, WG_SCAN = 2, WG_FETCH_GRAM = 3, WG_COMPACT = 4 };

struct WgLds { int L, D, xl, zl, lamA, lamB, pend_lam, u, mnew, gram, red, cand, dbl; int ws, slot, slot_id, freestk, pend_id, sense, cmd, ints; int total_bytes; };
__host__ __device__ inline int wg_round_up(int a, int b) { return (a + b - 1) / b * b; }
__host__ __device__ inline WgLds wg_lds(int n, int m, int cap, int capL, int W)
{
    WgLds s;
    const int cp = wg_round_up(cap, 2);
    int o = 0;
    s.L = o; o += wg_round_up(capL * (capL + 1) / 2, 2);
    s.D = o; o += cp; s.xl = o; o += cp; s.zl = o; o += cp; s.lamA = o; o += cp; s.lamB = o; o += cp; s.pend_lam = o; o += cp;
    s.u = o; o += wg_round_up(n, 2) + 2;
    s.mnew = o; o += wg_round_up(n, 2) + 2;
    s.gram = o; o += cp;
    s.red = o; o += 64 * W;            // partial sums of the split primal / Gram passes (W waves x 64 lanes)
    s.cand = o; o += 2 * W;            // per-wave scan candidates: value | (index, side)
    s.dbl = o;
    int oi = 0;
    s.ws = oi; oi += wg_round_up(cap, 4);
    s.slot = oi; oi += wg_round_up(cap, 4);
    s.slot_id = oi; oi += wg_round_up(cap, 4);
    s.freestk = oi; oi += wg_round_up(cap, 4);
    s.pend_id = oi; oi += wg_round_up(cap, 4);
    s.sense = oi; oi += wg_round_up(m, 4);
    s.cmd = oi; oi += 16;
    s.ints = oi;
    s.total_bytes = o * 8 + oi * 4;
    return s;
}

// what every thread of the workgroup knows (pointers and sizes; no iterate state)
struct WgCtx {
    int n, m, ms, cap, capL, npair, nblk, ldr, capT, W, exact;
    double *L, *D, *xl, *zl, *lamA, *lamB, *pend_lam, *u, *mnew, *gram, *red, *cand;
    int *ws, *slot, *slot_id, *freestk, *pend_id, *sense, *cmd;
    double *rowc, *rowcT;                 // this workgroup's scratch in HBM: [cap][ldr] and [n][capT]
    const double *Mblk, *dupper, *dlower, *scaling;
};

// the master's iterate
template <int C>
struct WgWave {
    WgCtx c;
    double *lam, *lams;
    int na, reuse, sing, has_soft, nfree, hi_slot, overflow;
    double fval, soft;
    DAQPSettings st;
    int *trace; int trace_cap, trace_len;
    unsigned long long t_start;
    long long prof[8]; bool profiling;
};
#define WPROF_T0(w) long long wprof_t0_ = (w).profiling ? (long long)__builtin_readcyclecounter() : 0
#define WPROF_ACC(w, slot) do { if ((w).profiling) { const long long t1_ = (long long)__builtin_readcyclecounter(); (w).prof[slot] += t1_ - wprof_t0_; wprof_t0_ = t1_; } } while (0)

template <int C>
__device__ __forceinline__ void wtrace(WgWave<C> &w, int ev)
{
    if (w.trace) {
        if (lane_id() == 0 && w.trace_len < w.trace_cap) w.trace[w.trace_len] = ev;
        w.trace_len++;
    }
}

__device__ __forceinline__ int wg_wave() { return (int)(threadIdx.x >> 6); }
__device__ __forceinline__ int wg_lane() { return (int)(threadIdx.x & 63); }

// ---------------------------------------------------------------------------------------------------------------------
// the parallel phases: executed by EVERY wave of the workgroup with the same arguments (barriers inside are workgroup-wide)
// ---------------------------------------------------------------------------------------------------------------------

// row `id` of the LDP constraint matrix -> LDS (mnew) and both orientations of the scratch at `slot`
__device__ __forceinline__ void wg_fetch_row(const WgCtx &c, int id, int slot, bool to_lds)
{
    const int t = (int)threadIdx.x;
    if (t < c.npair) {
        const double2 *src = reinterpret_cast<const double2 *>(c.Mblk) + ((size_t)(id >> 6) * c.npair) * 64 + (id & 63);
        const double2 v = src[(size_t)t * 64];
        const bool two = 2 * t + 1 < c.n;
        if (to_lds) { c.mnew[2 * t] = v.x; c.mnew[2 * t + 1] = two ? v.y : 0.0; }
        double *rw = c.rowc + (size_t)slot * c.ldr + 2 * t;
        rw[0] = v.x;
        if (two) rw[1] = v.y;
        c.rowcT[(size_t)(2 * t) * c.capT + slot] = v.x;
        if (two) c.rowcT[(size_t)(2 * t + 1) * c.capT + slot] = v.y;
    }
}

// u = -sum_i lam*_i row(ws[i])  (auxiliary.c:46-88): lane <-> column, rows in working-set order.  Exact mode: one sweep per
// column block; otherwise the working set is cut into contiguous segments, one per wave, and the partial sums are added
// in segment order.
__device__ __forceinline__ void wg_primal(const WgCtx &c, int na, const double *lams)
{
    const int wv = wg_wave(), lane = wg_lane();
    const int CB = (c.n + 63) >> 6;
    int segs = c.exact ? 1 : c.W / CB;
    if (segs < 1) segs = 1;
    if (segs > 4) segs = 4;
    const int cb = wv % CB, seg = wv / CB;
    const int per = (na + segs - 1) / segs;
    const int i0 = seg * per, i1 = (i0 + per < na) ? i0 + per : na;
    const int j = cb * 64 + lane;
    const int jj = j < c.n ? j : 0;
    double acc = 0;
    if (seg < segs) {
        for (int i = i0; i < i1; i += 8) {
            double rv[8], li[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int ii = (i + q < i1) ? i + q : i1 - 1;
                rv[q] = c.rowc[(size_t)c.slot[ii] * c.ldr + jj];
                li[q] = lams[ii];
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) if (i + q < i1) acc -= rv[q] * li[q];
        }
    }
    if (segs == 1) {
        if (wv < CB && j < c.n) c.u[j] = acc;
    } else {
        if (seg < segs) c.red[seg * 64 * CB + cb * 64 + lane] = acc;    // segs * CB <= W waves of 64 lanes
        __syncthreads();
        const int t = (int)threadIdx.x;
        if (t < c.n) {
            double s = c.red[t];
            for (int g = 1; g < segs; ++g) s += c.red[g * 64 * CB + t];
            c.u[t] = s;
        }
    }
}

// feasibility scan + most-violated pick (auxiliary.c:89-198): lane <-> constraint row, wave w takes the 64-row blocks
// w, w + W, ...; M streams from HBM, DEPTH x 16 bytes per lane in flight ahead of the k-ordered chain.  Every wave leaves
// its candidate (value, row, side) in LDS; the master picks among them (lowest value, then lowest row).
__device__ __forceinline__ void wg_scan(const WgCtx &c, double primal_tol)
{
    const int wv = wg_wave(), lane = wg_lane(), n = c.n;
    const double ep = -primal_tol;
    double bv = 0.0;
    int bi = kBig, bup = 0;
    const double2 *u2 = reinterpret_cast<const double2 *>(c.u);
    const bool odd = (n & 1) != 0;
    const int full = odd ? c.npair - 1 : c.npair;
    constexpr int DEPTH = 16;
    for (int blk = wv; blk < c.nblk; blk += c.W) {
        const int r = blk * 64 + lane;
        const bool own = r < c.m;
        const int rr = own ? r : 0;
        const double2 *src = reinterpret_cast<const double2 *>(c.Mblk) + ((size_t)blk * c.npair) * 64 + lane;
        // the row's bounds ride in the first batch of loads
        const double du = c.dupper[rr], dl = c.dlower[rr], sc = c.scaling[rr];
        double mu = 0;
        int t = 0;
        for (; t + DEPTH <= full; t += DEPTH) {
            double2 mm[DEPTH], uk[DEPTH];
#pragma unroll
            for (int q = 0; q < DEPTH; ++q) mm[q] = src[(size_t)(t + q) * 64];
#pragma unroll
            for (int q = 0; q < DEPTH; ++q) uk[q] = u2[t + q];
#pragma unroll
            for (int q = 0; q < DEPTH; ++q) { mu += mm[q].x * uk[q].x; mu += mm[q].y * uk[q].y; }
        }
        if (t < full) {
            double2 mm[DEPTH], uk[DEPTH];
#pragma unroll
            for (int q = 0; q < DEPTH; ++q) { const int tt = (t + q < full) ? t + q : full - 1; mm[q] = src[(size_t)tt * 64]; uk[q] = u2[tt]; }
#pragma unroll
            for (int q = 0; q < DEPTH; ++q) if (t + q < full) { mu += mm[q].x * uk[q].x; mu += mm[q].y * uk[q].y; }
        }
        if (odd) mu += src[(size_t)full * 64].x * c.u[n - 1];
        if (own) {
            const int sn = c.sense[r];
            if (!(sn & (DAQP_ACTIVE + DAQP_IMMUTABLE))) {
                const double bound = ep * sc;
                double cand = du - mu;
                if (cand < bv && cand < bound) { bv = cand; bi = r; bup = 1; }
                else {
                    cand = mu - dl;
                    if (cand < bv && cand < bound) { bv = cand; bi = r; bup = 0; }
                }
            }
        }
    }
    wave_argmin(bv, bi, bup);
    if (lane == 0) {
        c.cand[2 * wv] = bv;
        reinterpret_cast<int *>(c.cand + 2 * wv + 1)[0] = bi;
        reinterpret_cast<int *>(c.cand + 2 * wv + 1)[1] = bup;
    }
}

// Gram column of the LDL' append (factorization.c:21-60): gram[s] = row(slot s) . m_new for every slot up to `hi`
// (free slots hold stale rows: finite, unused).  lane <-> slot, rows read from the transposed scratch (coalesced).
// Exact mode: the reference's dot_row (four interleaved partial sums from the row's start column, (s0+s1)+(s2+s3));
// otherwise the columns are cut into segments, one per wave, combined in segment order.
__device__ __forceinline__ void wg_gram(const WgCtx &c, int id, int hi)
{
    const int wv = wg_wave(), lane = wg_lane(), n = c.n;
    const int RG = (hi + 63) >> 6;                       // row groups of 64 slots
    int segs = c.exact ? 1 : c.W / RG;
    if (segs < 1) segs = 1;
    const int rg = wv % RG, seg = wv / RG;
    const int s = rg * 64 + lane;
    const int ss = s < hi ? s : 0;
    const double *col = c.rowcT + ss;
    if (c.exact) {
        if (wv < RG) {
            const int c0 = id < c.ms ? id : 0;
            const int idk = c.slot_id[ss];
            const int j0 = (idk < c.ms) ? (c0 > idk ? c0 : idk) : c0;       // factorization.c:64-72
            const int len = n - j0, body = j0 + (len & ~3);
            double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
            for (int jb = 0; jb < n; jb += 8) {
                double rv[8], mv[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) { const int j = (jb + q < n) ? jb + q : n - 1; rv[q] = col[(size_t)j * c.capT]; mv[q] = c.mnew[j]; }
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const int j = jb + q;
                    if (j < n && j >= j0) {
                        const double p = rv[q] * mv[q];
                        const int ch = (j < body) ? ((j - j0) & 3) : 0;
                        if (ch == 0) s0 += p; else if (ch == 1) s1 += p; else if (ch == 2) s2 += p; else s3 += p;
                    }
                }
            }
            if (s < hi) c.gram[s] = (s0 + s1) + (s2 + s3);
        }
        return;
    }
    const int per = ((n + segs - 1) / segs + 7) & ~7;
    const int ja = seg * per, jb_ = (ja + per < n) ? ja + per : n;
    double acc = 0;
    if (seg < segs) {
        for (int j = ja; j < jb_; j += 8) {
            double rv[8], mv[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) { const int jj = (j + q < jb_) ? j + q : jb_ - 1; rv[q] = col[(size_t)jj * c.capT]; mv[q] = c.mnew[jj]; }
#pragma unroll
            for (int q = 0; q < 8; ++q) if (j + q < jb_) acc += rv[q] * mv[q];
        }
    }
    if (segs == 1) {
        if (wv < RG && s < hi) c.gram[s] = acc;
    } else {
        if (seg < segs) c.red[seg * 64 * RG + rg * 64 + lane] = acc;     // segs * RG <= W
        __syncthreads();
        const int t = (int)threadIdx.x;
        if (t < hi) {
            double g = c.red[t];
            for (int q = 1; q < segs; ++q) g += c.red[q * 64 * RG + t];
            c.gram[t] = g;
        }
    }
}

// factorization.c:121-129: move rows r+1.. of packed L up by one and drop column r, element-parallel over the
// destination range [tri(r), tri(na-1)).  A destination always reads from a higher address, so ascending chunks with
// "everybody reads, barrier, everybody writes, barrier" never clobber a live source.
__device__ __forceinline__ void wg_compact(const WgCtx &c, int r, int na)
{
    const int e0 = tri(r), e1 = tri(na - 1);
    const int T = 64 * c.W;
    constexpr int U = 4;
    for (int cb = e0; cb < e1; cb += T * U) {
        double tmp[U];
#pragma unroll
        for (int q = 0; q < U; ++q) {
            const int e = cb + q * T + (int)threadIdx.x;
            tmp[q] = 0;
            if (e < e1) {
                int i = (int)((sqrtf(8.0f * (float)e + 1.0f) - 1.0f) * 0.5f);
                while (tri(i + 1) <= e) ++i;
                while (tri(i) > e) --i;
                const int j = e - tri(i);
                tmp[q] = c.L[tri(i + 1) + j + (j >= r ? 1 : 0)];
            }
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < U; ++q) {
            const int e = cb + q * T + (int)threadIdx.x;
            if (e < e1) c.L[e] = tmp[q];
        }
        __syncthreads();
    }
}

// one command, executed by every wave (the master included)
__device__ __forceinline__ void wg_do(const WgCtx &c, int code, double primal_tol)
{
    const int a0 = c.cmd[1], a1 = c.cmd[2], na = c.cmd[3], hi = c.cmd[4];
    const double *lams = c.cmd[5] ? c.lamB : c.lamA;
    if (code == WG_PRIMAL) wg_primal(c, na, lams);
    else if (code == WG_SCAN) wg_scan(c, primal_tol);
    else if (code == WG_FETCH_GRAM) {
        wg_fetch_row(c, a0, a1, true);
        __syncthreads();
        wg_gram(c, a0, hi);
    } else if (code == WG_COMPACT) wg_compact(c, a0, na);
}

// master side: post a command, take part in it
template <int C>
__device__ __forceinline__ void wg_run(WgWave<C> &w, int code, int a0 = 0, int a1 = 0)
{
    const WgCtx &c = w.c;
    if (lane_id() == 0) {
        c.cmd[0] = code; c.cmd[1] = a0; c.cmd[2] = a1; c.cmd[3] = w.na; c.cmd[4] = w.hi_slot + 1;
        c.cmd[5] = (w.lams == c.lamB) ? 1 : 0;
    }
    __syncthreads();
    wg_do(c, code, w.st.primal_tol);
    __syncthreads();
}
// everybody else: serve commands until the master says EXIT
__device__ __forceinline__ void wg_serve(const WgCtx &c, double primal_tol)
{
    for (;;) {
        __syncthreads();
        const int code = c.cmd[0];
        if (code == WG_EXIT) break;
        wg_do(c, code, primal_tol);
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// master: the serial part of the iteration on wave 0 (lane + 64 c <-> working-set position), L and the vectors in LDS
// ---------------------------------------------------------------------------------------------------------------------
constexpr int kWPre = 8;

// LDL' row append (factorization.c:21-111)
template <int C>
__device__ __forceinline__ void wldl_append(WgWave<C> &w, int id)
{
    const WgCtx &c = w.c;
    const int lane = lane_id(), na = w.na, n = c.n, base = tri(na);
    if (na >= c.capL) { w.overflow = 1; return; }          // packed L would outgrow its LDS: the one-wave kernel takes this problem
    // a free slot of the active-row scratch (lowest first: the slots in use stay dense)
    const int newslot = c.freestk[w.nfree - 1];
    w.nfree--;
    if (newslot > w.hi_slot) w.hi_slot = newslot;
    if (lane == 0) { c.slot[na] = newslot; c.slot_id[newslot] = id; }
    wg_run(w, WG_FETCH_GRAM, id, newslot);
    w.sing = kEmpty;
    double g[C];
    int ns_act = 0;
#pragma unroll
    for (int cc = 0; cc < C; ++cc) {
        const int k = lane + 64 * cc;
        g[cc] = 0;
        int soft_k = 0;
        if (k <= na) {
            g[cc] = c.gram[c.slot[k]];
            const int idk = (k < na) ? c.ws[k] : id;
            soft_k = (c.sense[idk] & DAQP_SOFT) ? 1 : 0;
        }
        if (w.has_soft) ns_act += __popcll(__ballot(soft_k));
    }
    double dnew = rlc<C>(g, na);
    if (c.sense[id] & DAQP_SOFT) dnew += w.st.rho_soft;
    if (na == 0) {
        if (lane == 0) c.D[0] = dnew;
        WSYNC();
        return;
    }
    // forward substitution  l <- L \ g   (factorization.c:81-88)
    for (int j0 = 0; j0 < na - 1; j0 += kWPre) {
        double Lv[kWPre][C];
#pragma unroll
        for (int q = 0; q < kWPre; ++q)
#pragma unroll
            for (int cc = 0; cc < C; ++cc) {
                const int k = lane + 64 * cc, j = j0 + q;
                Lv[q][cc] = c.L[(k > j && k < na) ? tri(k) + j : 0];
            }
#pragma unroll
        for (int q = 0; q < kWPre; ++q) {
            const int j = j0 + q;
            if (j < na - 1) {
                const double lj = rlc<C>(g, j);
#pragma unroll
                for (int cc = 0; cc < C; ++cc) {
                    const int k = lane + 64 * cc;
                    if (k > j && k < na) g[cc] -= Lv[q][cc] * lj;
                }
            }
        }
    }
    double p[C];
#pragma unroll
    for (int cc = 0; cc < C; ++cc) {
        const int k = lane + 64 * cc;
        p[cc] = 0;
        if (k < na) {
            const double t = g[cc];
            const double lk = t / c.D[k];
            c.L[base + k] = lk;
            p[cc] = t * lk;
        }
    }
    double acc = dnew;
    for (int k = 0; k < na; ++k) acc -= rlc<C>(p, k);
    int sing = kEmpty;
    if (acc < w.st.sing_tol || na >= n + ns_act) { sing = na; acc = 0; }
    if (lane == 0) c.D[na] = acc;
    w.sing = sing;
    WSYNC();
}

// LDL' row delete (factorization.c:112-151)
template <int C>
__device__ __forceinline__ void wldl_delete(WgWave<C> &w, int r)
{
    const WgCtx &c = w.c;
    const int lane = lane_id(), na = w.na;
    if (na == r + 1) return;
    const int nupd = na - r - 1;
    double wv[C];
#pragma unroll
    for (int cc = 0; cc < C; ++cc) {
        const int t = lane + 64 * cc;
        wv[cc] = (t < nupd) ? c.L[tri(r + 1 + t) + r] : 0.0;
    }
    WSYNC();
    wg_run(w, WG_COMPACT, r);
    double alpha = c.D[r];
    for (int j0 = 0; j0 < nupd; j0 += kWPre) {
        // the lane's own L entries of kWPre columns (new numbering: row r+t, column r+j) before the chain, written back after
        double Lc[kWPre][C], Dv[kWPre];
#pragma unroll
        for (int q = 0; q < kWPre; ++q) {
            const int j = j0 + q;
            Dv[q] = c.D[(j < nupd) ? r + 1 + j : r];
#pragma unroll
            for (int cc = 0; cc < C; ++cc) {
                const int t = lane + 64 * cc;
                Lc[q][cc] = c.L[(t > j && t < nupd) ? tri(r + t) + r + j : 0];
            }
        }
#pragma unroll
        for (int q = 0; q < kWPre; ++q) {
            const int j = j0 + q;
            if (j < nupd) {
                const double p = rlc<C>(wv, j);
                const double Di = Dv[q];
                const double dbar = Di + alpha * p * p;
                const double beta = p * alpha / dbar;
                alpha = Di * alpha / dbar;
                if (lane == 0) c.D[r + j] = dbar;
#pragma unroll
                for (int cc = 0; cc < C; ++cc) {
                    const int t = lane + 64 * cc;
                    if (t > j && t < nupd) {
                        wv[cc] -= p * Lc[q][cc];
                        Lc[q][cc] = Lc[q][cc] + beta * wv[cc];
                    }
                }
            }
        }
#pragma unroll
        for (int q = 0; q < kWPre; ++q) {
            const int j = j0 + q;
#pragma unroll
            for (int cc = 0; cc < C; ++cc) {
                const int t = lane + 64 * cc;
                if (j < nupd && t > j && t < nupd) c.L[tri(r + t) + r + j] = Lc[q][cc];
            }
        }
    }
    WSYNC();
}

// auxiliary.c:3-22 without the trailing pivot; returns 1 if the factor became singular
template <int C>
__device__ __forceinline__ int wdrop_core(WgWave<C> &w, int r)
{
    const WgCtx &c = w.c;
    const int lane = lane_id();
    const int idr = c.ws[r];
    wtrace(w, -(idr + 1));
    if (lane == 0) { c.sense[idr] &= ~DAQP_ACTIVE; c.freestk[w.nfree] = c.slot[r]; }
    w.nfree++;
    wldl_delete(w, r);
    w.na--;
    int wsn[C], sln[C];
    double lmn[C];
#pragma unroll
    for (int cc = 0; cc < C; ++cc) {
        const int i = lane + 64 * cc;
        wsn[cc] = 0; sln[cc] = 0; lmn[cc] = 0;
        if (i >= r && i < w.na) { wsn[cc] = c.ws[i + 1]; sln[cc] = c.slot[i + 1]; lmn[cc] = w.lam[i + 1]; }
    }
    WSYNC();
#pragma unroll
    for (int cc = 0; cc < C; ++cc) {
        const int i = lane + 64 * cc;
        if (i >= r && i < w.na) { c.ws[i] = wsn[cc]; c.slot[i] = sln[cc]; w.lam[i] = lmn[cc]; }
    }
    if (r < w.reuse) w.reuse = r;
    int took = 0;
    WSYNC();
    if (w.na > 0 && c.D[w.na - 1] < w.st.sing_tol) {
        w.sing = w.na - 1;
        took = 1;
    }
    WSYNC();
    if (took && lane == 0) c.D[w.na - 1] = 0;
    WSYNC();
    return took;
}

template <int C>
__device__ __forceinline__ void wpush_core(WgWave<C> &w, int id, double lamv) // auxiliary.c:27-40
{
    const WgCtx &c = w.c;
    const int lane = lane_id();
    wtrace(w, id + 1);
    if (lane == 0) c.sense[id] |= DAQP_ACTIVE;
    WSYNC();
    wldl_append(w, id);
    if (w.overflow) return;
    if (lane == 0) { c.ws[w.na] = id; w.lam[w.na] = lamv; }
    w.na++;
    WSYNC();
}

// daqp_pivot_last (auxiliary.c:379-396) with its recursion as an explicit stack
template <int C>
__device__ __forceinline__ void wpivot_tail(WgWave<C> &w)
{
    const WgCtx &c = w.c;
    const int lane = lane_id();
    int depth = 0;
    for (;;) {
        if (w.overflow) break;
        const int r = w.na - 2;
        bool piv = false;
        if (w.na > 1) {
            const double dr = c.D[r], dl = c.D[w.na - 1];
            piv = dr < w.st.pivot_tol && dr < dl;
        }
        if (piv) {
            wtrace(w, kTracePivot);
            if (lane == 0) { c.pend_id[depth] = c.ws[r]; c.pend_lam[depth] = w.lam[r]; }
            depth++;
            WSYNC();
            if (wdrop_core(w, r)) break;
            continue;
        }
        if (depth == 0) break;
        if (w.sing != kEmpty) break;
        depth--;
        const int id = c.pend_id[depth];
        const double lv = c.pend_lam[depth];
        wpush_core(w, id, lv);
    }
}
template <int C>
__device__ __forceinline__ void wremove_constraint(WgWave<C> &w, int r) { if (!wdrop_core(w, r)) wpivot_tail(w); }
template <int C>
__device__ __forceinline__ void wadd_constraint(WgWave<C> &w, int id, double lamv) { wpush_core(w, id, lamv); if (!w.overflow) wpivot_tail(w); }

// b <- L' \ b for the leading cnt rows (product order b_j * L[j][i])
template <int C>
__device__ __forceinline__ void wbackward(WgWave<C> &w, double (&b)[C], int cnt)
{
    const WgCtx &c = w.c;
    const int lane = lane_id();
    for (int j0 = cnt - 1; j0 >= 1; j0 -= kWPre) {
        double Lv[kWPre][C];
#pragma unroll
        for (int q = 0; q < kWPre; ++q)
#pragma unroll
            for (int cc = 0; cc < C; ++cc) {
                const int i = lane + 64 * cc, j = j0 - q;
                Lv[q][cc] = c.L[(j >= 1 && i < j) ? tri(j) + i : 0];
            }
#pragma unroll
        for (int q = 0; q < kWPre; ++q) {
            const int j = j0 - q;
            if (j >= 1) {
                const double bj = rlc<C>(b, j);
#pragma unroll
                for (int cc = 0; cc < C; ++cc) {
                    const int i = lane + 64 * cc;
                    if (i < j) b[cc] -= bj * Lv[q][cc];
                }
            }
        }
    }
}
// x_i = rhs_i - sum_{j<i} L[i][j] x_j for rows i >= from (j ascending); rows < from are final in xl
template <int C>
__device__ __forceinline__ void wforward(WgWave<C> &w, double (&acc)[C], int from)
{
    const WgCtx &c = w.c;
    const int lane = lane_id(), na = w.na;
    if (from == na - 1 && na > 1) {
        // the usual case after an add: only the last row is open -- its products in parallel, then the j-ordered chain of
        // subtractions (the same operations, in the same order, as the sweep below performs for that row)
        double p[C];
#pragma unroll
        for (int cc = 0; cc < C; ++cc) {
            const int j = lane + 64 * cc;
            p[cc] = (j < na - 1) ? c.L[tri(na - 1) + j] * c.xl[j] : 0.0;
        }
        double last = rlc<C>(acc, na - 1);
        for (int j = 0; j < na - 1; ++j) last -= rlc<C>(p, j);
#pragma unroll
        for (int cc = 0; cc < C; ++cc) if (lane + 64 * cc == na - 1) acc[cc] = last;
        return;
    }
    for (int j0 = 0; j0 < na - 1; j0 += kWPre) {
        double Lv[kWPre][C], xf[kWPre];
#pragma unroll
        for (int q = 0; q < kWPre; ++q) {
            const int j = j0 + q;
            xf[q] = c.xl[(j < from) ? j : 0];
#pragma unroll
            for (int cc = 0; cc < C; ++cc) {
                const int i = lane + 64 * cc;
                Lv[q][cc] = c.L[(i >= from && i > j && i < na) ? tri(i) + j : 0];
            }
        }
#pragma unroll
        for (int q = 0; q < kWPre; ++q) {
            const int j = j0 + q;
            if (j < na - 1) {
                const double xj = (j < from) ? xf[q] : rlc<C>(acc, j);
#pragma unroll
                for (int cc = 0; cc < C; ++cc) {
                    const int i = lane + 64 * cc;
                    if (i >= from && i > j && i < na) acc[cc] -= Lv[q][cc] * xj;
                }
            }
        }
    }
}

template <int C>
__device__ __forceinline__ void wsolve_csp(WgWave<C> &w) // auxiliary.c:314-354
{
    const WgCtx &c = w.c;
    const int lane = lane_id(), na = w.na, from = w.reuse;
    double acc[C];
#pragma unroll
    for (int cc = 0; cc < C; ++cc) {
        const int i = lane + 64 * cc;
        acc[cc] = 0;
        if (i >= from && i < na) {
            const int id = c.ws[i];
            acc[cc] = (c.sense[id] & DAQP_LOWER) ? -c.dlower[id] : -c.dupper[id];
        }
    }
    wforward<C>(w, acc, from);
    double b[C];
#pragma unroll
    for (int cc = 0; cc < C; ++cc) {
        const int i = lane + 64 * cc;
        b[cc] = 0;
        if (i < na) {
            if (i >= from) {
                c.xl[i] = acc[cc];
                b[cc] = acc[cc] / c.D[i];
                c.zl[i] = b[cc];
            } else b[cc] = c.zl[i];
        }
    }
    wbackward<C>(w, b, na);
#pragma unroll
    for (int cc = 0; cc < C; ++cc) {
        const int i = lane + 64 * cc;
        if (i < na) w.lams[i] = b[cc];
    }
    w.reuse = na;
    WSYNC();
}

template <int C>
__device__ __forceinline__ void wsingular_direction(WgWave<C> &w) // auxiliary.c:357-376
{
    const WgCtx &c = w.c;
    const int lane = lane_id(), s = w.sing, base = tri(s);
    double b[C];
#pragma unroll
    for (int cc = 0; cc < C; ++cc) {
        const int i = lane + 64 * cc;
        b[cc] = (i < s) ? -c.L[base + i] : 0.0;
    }
    wbackward<C>(w, b, s);
    const bool flip = (c.sense[c.ws[s]] & DAQP_LOWER) != 0;
#pragma unroll
    for (int cc = 0; cc < C; ++cc) {
        const int i = lane + 64 * cc;
        if (i <= s) {
            const double v = (i == s) ? 1.0 : b[cc];
            w.lams[i] = flip ? -v : v;
        }
    }
    WSYNC();
}

// auxiliary.c:277-311 (SOFT_WEIGHTS off)
template <int C>
__device__ __forceinline__ int wremove_blocking(WgWave<C> &w)
{
    const WgCtx &c = w.c;
    const int lane = lane_id(), na = w.na;
    const double dtol = w.st.dual_tol;
    const bool regular = (w.sing == kEmpty);
    double bv = DAQP_INF;
    int bi = kBig, aux = 0;
    double lm[C], ls[C];
#pragma unroll
    for (int cc = 0; cc < C; ++cc) {
        const int i = lane + 64 * cc;
        lm[cc] = 0; ls[cc] = 0;
        if (i < na) {
            lm[cc] = w.lam[i]; ls[cc] = w.lams[i];
            const int sn = c.sense[c.ws[i]];
            bool blocking = !(sn & DAQP_IMMUTABLE);
            if (sn & DAQP_LOWER) { if (ls[cc] < dtol) blocking = false; }
            else if (ls[cc] > -dtol) blocking = false;
            if (blocking) {
                const double cand = regular ? -lm[cc] / (ls[cc] - lm[cc]) : -lm[cc] / ls[cc];
                if (cand < bv) { bv = cand; bi = i; }
            }
        }
    }
    wave_argmin(bv, bi, aux);
    if (bi == kBig) return 0;
    const double alpha = bv;
#pragma unroll
    for (int cc = 0; cc < C; ++cc) {
        const int i = lane + 64 * cc;
        if (i < na) w.lam[i] = regular ? lm[cc] + alpha * (ls[cc] - lm[cc]) : lm[cc] + alpha * ls[cc];
    }
    w.sing = kEmpty;
    WSYNC();
    wremove_constraint(w, bi);
    return 1;
}

// primal step (all waves) + the soft part of the objective (auxiliary.c:46-88)
template <int C>
__device__ __forceinline__ void wprimal(WgWave<C> &w)
{
    const WgCtx &c = w.c;
    wg_run(w, WG_PRIMAL);
    double fv = 0;
    if (w.has_soft) {
        for (int i = 0; i < w.na; ++i)
            if (c.sense[c.ws[i]] & DAQP_SOFT) { const double li = w.lams[i]; fv += li * li; }
    }
    fv = fv * w.st.rho_soft;
    w.soft = fv;
}
template <int C>
__device__ __forceinline__ double wordered_norm2(WgWave<C> &w, double start)
{
    const WgCtx &c = w.c;
    double fv = start;
    for (int j0 = 0; j0 < c.n; j0 += 8) {
        double uj[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) uj[q] = c.u[(j0 + q < c.n) ? j0 + q : 0];
#pragma unroll
        for (int q = 0; q < 8; ++q) if (j0 + q < c.n) fv += uj[q] * uj[q];
    }
    return fv;
}
// scan (all waves), then the pick among the waves' candidates and, when asked, |u|^2 in index order
template <int C>
__device__ __forceinline__ int wscan(WgWave<C> &w, int &upper, bool with_fval)
{
    const WgCtx &c = w.c;
    wg_run(w, WG_SCAN);
    if (with_fval) w.fval = wordered_norm2(w, w.soft);
    double bv = 0.0;
    int bi = kBig, bup = 0;
    for (int k = 0; k < c.W; ++k) {
        const double v = c.cand[2 * k];
        const int i = reinterpret_cast<const int *>(c.cand + 2 * k + 1)[0], up = reinterpret_cast<const int *>(c.cand + 2 * k + 1)[1];
        if (i != kBig && (bi == kBig || v < bv || (v == bv && i < bi))) { bv = v; bi = i; bup = up; }
    }
    upper = bup;
    return bi;
}

template <int C>
__device__ __forceinline__ void wcommit_add(WgWave<C> &w, int pick, int upper) // auxiliary.c:152-166
{
    const WgCtx &c = w.c;
    if (lane_id() == 0) {
        if (upper) c.sense[pick] &= ~DAQP_LOWER; else c.sense[pick] |= DAQP_LOWER;
    }
    double *t = w.lam; w.lam = w.lams; w.lams = t;
    WSYNC();
    wadd_constraint(w, pick, upper ? 1.0 : -1.0);
}

// one step of iterative refinement on the active rows (auxiliary.c:498-593); rare, so the rows are read from the
// row-major scratch one lane per row
template <int C>
__device__ __forceinline__ void wrefine_active(WgWave<C> &w)
{
    const WgCtx &c = w.c;
    const int lane = lane_id(), na = w.na, n = c.n;
    w.reuse = 0;
    double acc[C];
#pragma unroll
    for (int cc = 0; cc < C; ++cc) {
        const int i = lane + 64 * cc;
        acc[cc] = 0;
        if (i < na) {
            const int id = c.ws[i];
            const double *row = c.rowc + (size_t)c.slot[i] * c.ldr;
            double mu = 0;
            for (int j = (id < c.ms ? id : 0); j < n; ++j) mu += row[j] * c.u[j];
            const double d = (c.sense[id] & DAQP_LOWER) ? c.dlower[id] : c.dupper[id];
            acc[cc] = mu - d;
            if (c.sense[id] & DAQP_SOFT) acc[cc] -= w.st.rho_soft * w.lams[i];
        }
    }
    wforward<C>(w, acc, 0);
    double b[C];
#pragma unroll
    for (int cc = 0; cc < C; ++cc) {
        const int i = lane + 64 * cc;
        b[cc] = 0;
        if (i < na) { c.xl[i] = acc[cc]; b[cc] = acc[cc] / c.D[i]; c.zl[i] = b[cc]; }
    }
    wbackward<C>(w, b, na);
#pragma unroll
    for (int cc = 0; cc < C; ++cc) {
        const int i = lane + 64 * cc;
        if (i < na) { c.xl[i] = b[cc]; w.lams[i] += b[cc]; }
    }
    WSYNC();
    double uu[C == 1 ? 2 : C + 1];   // columns: lane + 64 cc covers n <= 64 (C + 1) (n < cap <= 64 C)
    constexpr int CU = (C == 1 ? 2 : C + 1);
#pragma unroll
    for (int cc = 0; cc < CU; ++cc) { const int j = lane + 64 * cc; uu[cc] = (j < n) ? c.u[j] : 0.0; }
    for (int i = 0; i < na; ++i) {
        const double dl = c.xl[i];
        const int id = c.ws[i];
        const int j0 = id < c.ms ? id : 0;
        const double *row = c.rowc + (size_t)c.slot[i] * c.ldr;
#pragma unroll
        for (int cc = 0; cc < CU; ++cc) {
            const int j = lane + 64 * cc;
            if (j < n && j >= j0) uu[cc] -= row[j] * dl;
        }
    }
    WSYNC();
#pragma unroll
    for (int cc = 0; cc < CU; ++cc) { const int j = lane + 64 * cc; if (j < n) c.u[j] = uu[cc]; }
    WSYNC();
    w.fval = wordered_norm2(w, w.soft);
}

template <int C>
__device__ __forceinline__ void wreset_ws(WgWave<C> &w)
{
    const WgCtx &c = w.c;
    w.sing = kEmpty; w.na = 0; w.reuse = 0;
    // every slot of the scratch is free again, lowest on top
    for (int i = lane_id(); i < c.cap; i += 64) c.freestk[i] = c.cap - 1 - i;
    w.nfree = c.cap; w.hi_slot = -1;
    WSYNC();
}

// (re)build the working set from the ACTIVE bits, in index order (auxiliary.c:399-479)
template <int C>
__device__ __forceinline__ int wactivate_marked(WgWave<C> &w)
{
    const WgCtx &c = w.c;
    const int lane = lane_id();
    for (int blk = 0; blk * 64 < c.m; ++blk) {
        const int r = blk * 64 + lane;
        unsigned long long msk = __ballot(r < c.m && (c.sense[r] & DAQP_ACTIVE));
        while (msk) {
            const int i = blk * 64 + __ffsll((long long)msk) - 1;
            msk &= msk - 1;
            wadd_constraint(w, i, (c.sense[i] & DAQP_LOWER) ? -1.0 : 1.0);
            if (w.overflow) return 1;
            if (w.sing == kEmpty) continue;
            const int last = c.ws[w.na - 1];
            if (c.sense[last] & DAQP_IMMUTABLE) {
                wsingular_direction(w);
                double resid = 0.0, scale = 1.0;
                for (int j = 0; j < w.na; ++j) {
                    const int id = c.ws[j];
                    const double bd = (c.sense[id] & DAQP_LOWER) ? c.dlower[id] : c.dupper[id];
                    const double t = w.lams[j] * bd;
                    resid += t;
                    scale += t < 0 ? -t : t;
                }
                WSYNC();
                if (lane == 0) { c.sense[last] &= ~DAQP_ACTIVE; c.freestk[w.nfree] = c.slot[w.na - 1]; }
                w.nfree++;
                w.na--;
                w.sing = kEmpty;
                if (w.reuse > w.na) w.reuse = w.na;
                WSYNC();
                if (resid <= w.st.primal_tol * scale && resid >= -w.st.primal_tol * scale) continue;
                return DAQP_EXIT_OVERDETERMINED_INITIAL;
            }
            int flag = 1;
            for (int q = i; q < c.m; q += 1) {
                const int sn = c.sense[q];
                if (sn & DAQP_ACTIVE) {
                    if (sn & DAQP_IMMUTABLE) flag = DAQP_EXIT_OVERDETERMINED_INITIAL;
                    else if (lane == 0) c.sense[q] = sn & ~DAQP_ACTIVE;
                }
            }
            if (lane == 0) c.freestk[w.nfree] = c.slot[w.na - 1];
            w.nfree++;
            w.na--;
            w.sing = kEmpty;
            WSYNC();
            return flag;
        }
    }
    return 1;
}

// daqp_ldp (daqp.c:6-108)
template <int C>
__device__ __forceinline__ int wldp_loop(WgWave<C> &w, int &iterations)
{
    const WgCtx &c = w.c;
    const int lane = lane_id();
    int flag = DAQP_EXIT_ITERLIMIT, it, repaired = 0, stall = 0;
    double best = -1;
    const double fbound = 2 * w.st.fval_bound;
    const bool timed = w.st.time_limit > 0;
    for (it = 1; it < w.st.iter_limit; ++it) {
        if (w.overflow) break;
        WPROF_T0(w);
        if (w.sing == kEmpty) {
            wsolve_csp(w);
            WPROF_ACC(w, 0);
            const int blocked = wremove_blocking(w);
            if (blocked) WPROF_ACC(w, 5); else WPROF_ACC(w, 1);
            if (blocked) {   // falls through to the end of the reference's loop body: the clock check applies
                if (timed && (it & 31) == 0 && time_is_up(w.t_start, w.st.time_limit)) { flag = DAQP_EXIT_TIMELIMIT; break; }
                continue;
            }
            wprimal(w);
            WPROF_ACC(w, 2);
            int upper = 0;
            int pick = wscan(w, upper, true);
            WPROF_ACC(w, 3);
            if (w.fval > fbound) { flag = DAQP_EXIT_INFEASIBLE; break; }
            if (pick == kBig) {
                double dmin = c.D[0];
                for (int i = 1; i < w.na; ++i) { const double di = c.D[i]; if (di < dmin) dmin = di; }
                if (w.na > 2 && repaired != 1 && dmin < w.st.refactor_tol) {
                    repaired = 1;
                    wtrace(w, kTraceRefactor);
                    for (int i = lane; i < w.na; i += 64) {
                        const int id = c.ws[i];
                        if (w.lam[i] >= 0) c.sense[id] &= ~DAQP_LOWER; else c.sense[id] |= DAQP_LOWER;
                    }
                    WSYNC();
                    wreset_ws(w);
                    wactivate_marked(w);
                    continue;
                }
                if (w.na > 0 && dmin < w.st.pivot_tol) {
                    wtrace(w, kTraceRefine);
                    wrefine_active(w);
                    pick = wscan(w, upper, false);
                    if (pick != kBig) { wcommit_add(w, pick, upper); continue; }
                }
                flag = (w.soft > w.st.primal_tol) ? DAQP_EXIT_SOFT_OPTIMAL : DAQP_EXIT_OPTIMAL;
                break;
            }
            wcommit_add(w, pick, upper);
            WPROF_ACC(w, 4);
            if (w.fval - best < w.st.progress_tol) {
                if (stall++ > w.st.cycle_tol) {
                    if (repaired == 1) { flag = DAQP_EXIT_CYCLE; break; }
                    repaired = 1;
                    wtrace(w, kTraceCycleReset);
                    wreset_ws(w);
                    wactivate_marked(w);
                    stall = 0;
                    best = -1;
                }
            } else { best = w.fval; stall = 0; }
        } else {
            wtrace(w, kTraceSingular);
            wsingular_direction(w);
            if (!wremove_blocking(w)) { flag = DAQP_EXIT_INFEASIBLE; break; }
        }
        if (timed && (it & 31) == 0 && time_is_up(w.t_start, w.st.time_limit)) { flag = DAQP_EXIT_TIMELIMIT; break; }
    }
    iterations = it;
    return flag;
}

} // namespace daqp_amd
