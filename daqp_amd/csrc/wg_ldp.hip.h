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
// Round 6: where two four-wave workgroups fit a CU, one problem's serial master phases run under the other's bandwidth phases (1.5-1.6x on
// shapes whose factor fits half the LDS).  For the shapes whose factor does not (C4), the inverse factor of a cold solve is TIERED: rows below
// r0 in LDS, the rest in the problem's slot of the stored factor in HBM -- WROW() below; the kernel instantiation <C, false, true> of
// wg_kernel.hip.h.  Every matrix-vector product over W takes the HBM rows as one more batch of independent loads; nothing that walks a
// dependent chain over the factor (the L representation, the exact mode) runs on a tiered factor.
#pragma once
#include "wave_ldp.hip.h"
#include "wave_ldp_reg.hip.h"   // static_for
#include "wg_layout.hip.h"

namespace daqp_amd {

// 16-byte loads in flight per lane in the feasibility scans.  16, measured on C4 (profiles/r04d_scan_depth.txt): 21.2 k cycles per scan at 32,
// 18.2 k at 16, 19.4 k at 8 -- the stream runs at what a CU's miss path delivers either way (~30 B/clk), and 64 fewer registers in
// flight take the spills out of the phases around it
#ifndef DAQP_WG_SCAN_DEPTH
#define DAQP_WG_SCAN_DEPTH 16
#endif

enum : int { WG_EXIT = 0, WG_PRIMAL = 1, WG_SCAN = 2, WG_FETCH_GRAM = 3, WG_COMPACT = 4, WG_SCAN32 = 5, WG_WCSP = 6, WG_WAPPEND = 7, WG_WDELETE = 8, WG_W2L = 9 };

// LDS layout for working sets of up to 64*C rows.  Everything but the packed L sits at COMPILE-TIME offsets (vectors sized
// for 64*C rows, u / m_new for n <= 256): an LDS address is then "immediate + 8*index", and nothing about the layout has to
// be kept in registers across the master's state machine (twenty region pointers did not fit the scalar file and came back
// as VGPR-lane and scratch spills).  Only L, behind the m sense words, starts at a run-time offset.
template <int C>
struct WgL {
    static constexpr int CAP = 64 * C;
    // doubles
    static constexpr int D = 0, xl = CAP, zl = 2 * CAP, lamA = 3 * CAP, lamB = 4 * CAP, pend_lam = 5 * CAP, gram = 6 * CAP, rhs = 7 * CAP, u = 8 * CAP,
                         mnew = u + 258, red = mnew + 258, cand = red + 128 * kWgMaxWaves, prof = cand + 8 * kWgMaxWaves, dend = prof + 24;
    // ints, counted from double offset dend
    static constexpr int ws = 0, slot = CAP, slot_id = 2 * CAP, freestk = 3 * CAP, pend_id = 4 * CAP, cmd = 5 * CAP, sense = 5 * CAP + 16;
};
// what every thread of the workgroup knows (sizes, HBM pointers; no iterate state)
struct WgCtx {
    int n, m, ms, cap, capL, npair, nblk, ldr, capT, W, exact, oL, lmax;   // lmax: last valid index of packed L
    // Tiered inverse factor (k_ldp_wg<C, false, true>: two four-wave workgroups per CU): rows < r0 of W in LDS, rows >= r0 in the problem's own
    // slot of the stored factor in HBM (gL: element (i, j) at tri(i) + j -- the place where the row ends up anyway when the solve is over, and
    // L2-resident while the problem is in flight).  tier = 0 (a compile-time constant in every other instantiation): everything in LDS.
    int tier, r0, capW;                   // capW: rows the inverse factor may grow to (capL without tiers)
    double *gL;
    double *rowc, *rowcT;                 // this workgroup's scratch in HBM: [cap][ldr] and [n][capT]
    const double *Mblk, *dupper, *dlower, *scaling;
    const float *M32;                     // fp32 image of M for the screening scan (null: every scan in fp64)
    int nquad;
};
__device__ __forceinline__ double *wg_sm() { extern __shared__ __attribute__((aligned(16))) double wg_dyn_lds[]; return wg_dyn_lds; }
#define SD(c, name) (wg_sm() + WgL<C>::name)
#define SDL(c) (wg_sm() + (c).oL)
// row i of the (inverse) factor, tier-aware: a pointer to its element (i, 0).  In a tiered launch the pointer is generic (LDS or HBM by the row)
#define WROW(c, i) ((((c).tier && (i) >= (c).r0) ? (c).gL : SDL(c)) + tri(i))
// index into packed L clamped to its LDS allocation: lanes whose row does not exist load anyway (the value is discarded),
// so the address must stay inside the allocation
#define WLIDX(c, e) ((e) < (c).lmax ? (e) : (c).lmax)
#define SI(c, name) (reinterpret_cast<int *>(wg_sm() + WgL<C>::dend) + WgL<C>::name)
#define SI64(c) (reinterpret_cast<long long *>(wg_sm() + WgL<C>::prof) + 23)      // probe scratch: the cycle stamp of the current command's start
// probes inside a command (thread 0 of the workgroup; armed by cmd[6]): WGPROBE_BEGIN stamps, WGPROBE adds the time since the last stamp to a slot
#define WGPROBE_BEGIN(c) do { if (SI(c, cmd)[6] && wg_tid() == 0) SI64(c)[0] = (long long)__builtin_readcyclecounter(); } while (0)
#define WGPROBE(c, slot) do { if (SI(c, cmd)[6] && wg_tid() == 0) { const long long n_ = (long long)__builtin_readcyclecounter(); \
    reinterpret_cast<long long *>(wg_sm() + WgL<C>::prof)[slot] += n_ - SI64(c)[0]; SI64(c)[0] = n_; } } while (0)

// Thread coordinates as values the optimizer cannot see through.  Everything below runs inside two nested loops (the
// persistent workgroup's problem loop and the master's state machine); addresses of the form "base + f(lane)" are loop
// invariant, LICM hoists every one of them (tri(lane + 64 c), 8 * lane, row pointers ...) to the top of the kernel, the
// register file cannot hold them and they come back as scratch reloads -- a trip to memory to save a multiply.  Built from
// these, an address is recomputed where it is used.
__device__ __forceinline__ int wg_tid() { int t = (int)threadIdx.x; asm volatile("" : "+v"(t)); return t; }
__device__ __forceinline__ int wg_wave() { return __builtin_amdgcn_readfirstlane(wg_tid() >> 6); }
__device__ __forceinline__ int wg_lane() { return wg_tid() & 63; }
// Values that are the same in every lane but come out of LDS / HBM loads are VGPRs to the compiler: a branch on one makes
// the master's whole control flow "divergent" (exec-masked, with readfirstlane waterfalls around every v_readlane of the
// substitution chains).  Anything that steers control flow or indexes a cross-lane read goes through these first.
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ bool ub(bool b) { return __builtin_amdgcn_readfirstlane((int)b) != 0; }
__device__ __forceinline__ float rlf(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }
__device__ __forceinline__ double und(double v)
{
    return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)), __builtin_amdgcn_readfirstlane(__double2loint(v)));
}

// the master's iterate
template <int C>
struct WgWave {
    WgCtx c;
    int lam_b;                            // 1: lam lives in buffer B and lam* in buffer A (the reference swaps the two pointers on every add)
    int na, reuse, sing, has_soft, nfree, hi_slot, overflow;
    int use_w;                            // 1: the LDS factor area holds W = L^-1 (default arithmetic, regular factor), see "inverse factor" below
    int fast_na;                          // inverse factor: na right after a regular append (the next CSP is then an O(na) update, wdirection), else -1
    double fval, soft;
    const DAQPSettings *stp;              // device copy of the settings: scalar loads at the point of use
    int *trace; int trace_cap, trace_len;
    unsigned long long t_start;
    double tick_s;
    bool profiling;                       // phase cycle counters in LDS (WgL::prof)
};
#define WLAM(w) (wg_sm() + ((w).lam_b ? WgL<C>::lamB : WgL<C>::lamA))
#define WLAMS(w) (wg_sm() + ((w).lam_b ? WgL<C>::lamA : WgL<C>::lamB))
#define WPROF_T0(w) long long wprof_t0_ = (w).profiling ? (long long)__builtin_readcyclecounter() : 0
#define WPROF_ACC(w, slot) do { if ((w).profiling) { const long long t1_ = (long long)__builtin_readcyclecounter(); if (wg_lane() == 0) reinterpret_cast<long long *>(wg_sm() + WgL<C>::prof)[slot] += t1_ - wprof_t0_; wprof_t0_ = t1_; } } while (0)

template <int C>
__device__ __forceinline__ void wtrace(WgWave<C> &w, int ev)
{
    if (w.trace) {
        if (wg_lane() == 0 && w.trace_len < w.trace_cap) w.trace[w.trace_len] = ev;
        w.trace_len++;
    }
}


// ---------------------------------------------------------------------------------------------------------------------
// the parallel phases: executed by EVERY wave of the workgroup with the same arguments (barriers inside are workgroup-wide)
// ---------------------------------------------------------------------------------------------------------------------

// one column pair of an active row into the scratch: row-major [slot][ldr] (ldr: a multiple of 32 doubles, 16-byte stores, the pad
// columns n .. ldr-1 are zero from the allocation on and stay zero); the transposed copy [n][capT] is kept in the exact mode only --
// its Gram column follows the reference's dot_row per slot (lane <-> slot); the default mode reads the row-major copy alone
__device__ __forceinline__ void wg_store_row_pair(const WgCtx &c, int slot, int t, double2 v, bool two)
{
    if (!two) v.y = 0.0;
    *reinterpret_cast<double2 *>(c.rowc + (size_t)slot * c.ldr + 2 * t) = v;
    if (c.exact) {
        c.rowcT[(size_t)(2 * t) * c.capT + slot] = v.x;
        if (two) c.rowcT[(size_t)(2 * t + 1) * c.capT + slot] = v.y;
    }
}

// row `id` of the LDP constraint matrix -> LDS (mnew) and the scratch at `slot`
template <int C>
__device__ __forceinline__ void wg_fetch_row(const WgCtx &c, int id, int slot, bool to_lds)
{
    const int t = wg_tid();
    if (t < c.npair) {
        const double2 *src = reinterpret_cast<const double2 *>(c.Mblk) + ((size_t)(id >> 6) * c.npair) * 64 + (id & 63);
        const double2 v = src[(size_t)t * 64];
        const bool two = 2 * t + 1 < c.n;
        if (to_lds) { SD(c, mnew)[2 * t] = v.x; SD(c, mnew)[2 * t + 1] = two ? v.y : 0.0; }
        wg_store_row_pair(c, slot, t, v, two);
    }
    if (to_lds && t >= c.npair && 2 * t < 258) { SD(c, mnew)[2 * t] = 0.0; if (2 * t + 1 < 258) SD(c, mnew)[2 * t + 1] = 0.0; }   // the Gram pass reads whole 32-column chunks
}

// u = -sum_i lam*_i row(ws[i])  (auxiliary.c:46-88): lane <-> column, rows in working-set order.  Exact mode: one sweep per
// column block; otherwise the working set is cut into contiguous segments, one per wave, and the partial sums are added
// in segment order.
template <int C>
__device__ __forceinline__ void wg_primal(const WgCtx &c, int na, const double *lams)
{
    const int wv = wg_wave(), lane = wg_lane();
    if (!c.exact) {
        // default mode: lane <-> column PAIR (16-byte loads: twice the bytes in flight per lane -- these passes run at
        // bytes in flight / latency, and the latency is set by the whole chip streaming at once), the working set cut into
        // W / (column blocks) segments, partial sums added in segment order
        const int hp = c.ldr >> 1;                      // pairs per row, pad included (pad columns are zero)
        const int CB = (c.npair + 63) >> 6;
        int segs = c.W / CB;
        if (segs < 1) segs = 1;
        const int cb = wv % CB, seg = wv / CB;
        const int per = (na + segs - 1) / segs;
        const int i0 = seg * per, i1 = (i0 + per < na) ? i0 + per : na;
        const int pj = cb * 64 + lane;
        const int pp = pj < hp ? pj : 0;
        double ax = 0, ay = 0;
        if (seg < segs) {
            constexpr int PB = 24;     // rows in flight per lane
            for (int i = i0; i < i1; i += PB) {
                double2 rv[PB];
                double li[PB];
#pragma unroll
                for (int q = 0; q < PB; ++q) {
                    const int ii = (i + q < i1) ? i + q : i1 - 1;
                    rv[q] = reinterpret_cast<const double2 *>(c.rowc + (size_t)SI(c, slot)[ii] * c.ldr)[pp];
                    li[q] = (i + q < i1) ? lams[ii] : 0.0;
                }
#pragma unroll
                for (int q = 0; q < PB; ++q) { ax = __builtin_fma(-rv[q].x, li[q], ax); ay = __builtin_fma(-rv[q].y, li[q], ay); }
            }
        }
        double2 *red2 = reinterpret_cast<double2 *>(SD(c, red));
        if (seg < segs) { double2 a; a.x = ax; a.y = ay; red2[seg * 64 * CB + cb * 64 + lane] = a; }    // segs * CB <= W waves of 64 lanes, 16 bytes each
        __syncthreads();
        const int t = wg_tid();
        if (t < c.npair) {
            double2 sacc = red2[t];
            for (int g = 1; g < segs; ++g) { const double2 o = red2[g * 64 * CB + t]; sacc.x += o.x; sacc.y += o.y; }
            SD(c, u)[2 * t] = sacc.x;
            if (2 * t + 1 < c.n) SD(c, u)[2 * t + 1] = sacc.y;
        }
        return;
    }
    const int CB = (c.n + 63) >> 6;
    const int cb = wv % CB;
    const int j = cb * 64 + lane;
    const int jj = j < c.n ? j : 0;
    double acc = 0;
    if (wv < CB) {     // exact mode: one sweep per column block, rows in working-set order (auxiliary.c:46-88)
        constexpr int PB = 24;
        for (int i = 0; i < na; i += PB) {
            double rv[PB], li[PB];
#pragma unroll
            for (int q = 0; q < PB; ++q) {
                const int ii = (i + q < na) ? i + q : na - 1;
                rv[q] = c.rowc[(size_t)SI(c, slot)[ii] * c.ldr + jj];
                li[q] = (i + q < na) ? lams[ii] : 0.0;
            }
#pragma unroll
            for (int q = 0; q < PB; ++q) if (i + q < na) acc -= rv[q] * li[q];
        }
        if (j < c.n) SD(c, u)[j] = acc;
    }
}

// feasibility scan + most-violated pick (auxiliary.c:89-198): lane <-> constraint row, wave w takes the 64-row blocks
// w, w + W, ...; M streams from HBM, DEPTH x 16 bytes per lane in flight ahead of the k-ordered chain.  Every wave leaves
// its candidate (value, row, side) in LDS; the master picks among them (lowest value, then lowest row).
template <int C>
__device__ __forceinline__ void wg_scan(const WgCtx &c, double primal_tol)
{
    const int wv = wg_wave(), lane = wg_lane(), n = c.n;
    const double ep = -primal_tol;
    double bv = 0.0;
    int bi = kBig, bup = 0;
    const double2 *u2 = reinterpret_cast<const double2 *>(SD(c, u));
    const bool odd = (n & 1) != 0;
    const int full = odd ? c.npair - 1 : c.npair;
    constexpr int DEPTH = DAQP_WG_SCAN_DEPTH;   // 16-byte loads in flight per lane
    // candidate of row r given mu = M_r . u (auxiliary.c:126-150)
    auto consider = [&](int r, double mu, double du, double dl, double sc) __attribute__((always_inline)) {
        const int sn = SI(c, sense)[r];
        if (!(sn & (DAQP_ACTIVE + DAQP_IMMUTABLE))) {
            const double bound = ep * sc;
            double cand = du - mu;
            if (cand < bv && cand < bound) { bv = cand; bi = r; bup = 1; }
            else {
                cand = mu - dl;
                if (cand < bv && cand < bound) { bv = cand; bi = r; bup = 0; }
            }
        }
    };
    // one 64-row block per wave and trip, each lane its row's k-ordered chain.  (Measured on C4, 10 blocks over 8 waves: sharing
    // the two left-over blocks between all waves -- k-slices, partial sums through LDS -- is slower, 40 k instead of 31 k cycles
    // per scan: the scan runs at what the CU can pull from L2 / MALL, not at the waves' load depth.)
    for (int blk = (wv + c.W - 1) % c.W; blk < c.nblk; blk += c.W) {   // (wave 0, the master, takes its blocks last)
        const int r = blk * 64 + lane;
        const bool own = r < c.m;
        const int rr = own ? r : 0;
        const double2 *src = reinterpret_cast<const double2 *>(c.Mblk) + ((size_t)blk * c.npair) * 64 + lane;
        // the row's bounds ride in the first batch of loads
        const double du = c.dupper[rr], dl = c.dlower[rr], sc = c.scaling[rr];
        double mu = 0;
        int t = 0;
        for (; t + DEPTH <= full; t += DEPTH) {
            double2 mm[DEPTH];
#pragma unroll
            for (int q = 0; q < DEPTH; ++q) mm[q] = src[(size_t)(t + q) * 64];
#pragma unroll
            for (int g = 0; g < DEPTH / 8; ++g) {      // u from LDS eight pairs at a time (broadcast reads), k-ordered chain
                double2 uk[8];
#pragma unroll
                for (int h = 0; h < 8; ++h) uk[h] = u2[t + 8 * g + h];
#pragma unroll
                for (int h = 0; h < 8; ++h) { mu += mm[8 * g + h].x * uk[h].x; mu += mm[8 * g + h].y * uk[h].y; }
            }
        }
        if (t < full) {
            double2 mm[DEPTH];
#pragma unroll
            for (int q = 0; q < DEPTH; ++q) { const int tt = (t + q < full) ? t + q : full - 1; mm[q] = src[(size_t)tt * 64]; }
#pragma unroll
            for (int g = 0; g < DEPTH / 8; ++g) {
                double2 uk[8];
#pragma unroll
                for (int h = 0; h < 8; ++h) { const int tt = (t + 8 * g + h < full) ? t + 8 * g + h : full - 1; uk[h] = u2[tt]; }
#pragma unroll
                for (int h = 0; h < 8; ++h) if (t + 8 * g + h < full) { mu += mm[8 * g + h].x * uk[h].x; mu += mm[8 * g + h].y * uk[h].y; }
            }
        }
        if (odd) mu += src[(size_t)full * 64].x * SD(c, u)[n - 1];
        if (own) consider(r, mu, du, dl, sc);
    }
    wave_argmin(bv, bi, bup);
    if (lane == 0) {
        SD(c, cand)[8 * wv] = bv;
        reinterpret_cast<int *>(SD(c, cand) + 8 * wv + 1)[0] = bi;
        reinterpret_cast<int *>(SD(c, cand) + 8 * wv + 1)[1] = bup;
    }
}

// Screening scan in fp32 (default arithmetic mode): the same pass over an fp32 image of M, half the bytes.  It does not decide
// anything the fp64 scan would decide differently: with E a rigorous bound on |M_r.u (fp32) - M_r.u (exact)| (unit rows:
// E = (n/4 + 8) 2^-24 |u|), the master accepts its pick only if that row is the most violated one by more than 2E, violates
// its own threshold by more than E and its side is unambiguous, accepts "nothing violated" only if every row clears its
// threshold by more than E, and otherwise runs the fp64 scan above.  Each wave leaves in LDS: its best value s1 (as the
// fp64 scan's candidate value, du - mu or mu - dl), row, side, the runner-up value s2, the smallest margin to the threshold
// over its rows, the winner's own margin and side gap, and whether anything was not finite.
#ifndef DAQP_WG_SCAN_U_REGS
#define DAQP_WG_SCAN_U_REGS 1
#endif
#if DAQP_WG_SCAN_U_REGS
// u is read as FOUR fp32 registers per lane (column lane + 64 i; zero from n on) and a product's u operand is a scalar read out of a
// register (v_readlane with a constant lane), not a broadcast read of LDS: no fp32 copy of u to publish, no barrier before the stream,
// and no second set of 16-byte temporaries next to the loads in flight (the LDS reads of a batch were hoisted above its arithmetic:
// as many registers again as the batch itself).  Every load is "uniform base + lane": the block's start and the quad's offset are
// scalars (and told so with readfirstlane: inside the loops the addresses otherwise become sixteen per-lane pointer induction
// variables), the lane's 16 bytes a 32-bit offset shared by all loads of the pass.
template <int C>
__device__ __forceinline__ void wg_scan32(const WgCtx &c, double primal_tol)
{
    const int wv = wg_wave(), lane = wg_lane(), n = c.n;
    const unsigned lane_u = (unsigned)lane;
    constexpr int DEPTH = 16;                               // 16-byte loads in flight per lane = the 64 columns one register of u covers
    const int blk0 = (wv + c.W - 1) % c.W;                  // (wave 0, the master, takes its blocks last)
    float4 mm[DEPTH];
    const float4 *M4 = reinterpret_cast<const float4 *>(c.M32);
    auto load_batch = [&](int blk, int t) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < DEPTH; ++q) {
            const int tt = (t + q < c.nquad) ? t + q : c.nquad - 1;
            const int off = uni((blk * c.nquad + tt) * 64);
            mm[q] = (M4 + off)[lane_u];
        }
    };
    // Whole 64-row blocks, one per wave and round, as far as they come out even (nfull = the largest multiple of W); the blocks left
    // over -- two of C4's ten -- are cut by 64-column batches over ALL waves (DAQP_WG_SCAN_SPLIT): a wave streams at ~6 B/clk whatever
    // its depth, so a scan lasts as long as its busiest wave, and with two waves on a second block the other six had nothing in flight
    // for the second half of it.  The batches' partial sums meet in LDS (red), in batch order.
#ifndef DAQP_WG_SCAN_SPLIT
#define DAQP_WG_SCAN_SPLIT 1
#endif
    const int nb = (c.nquad + DEPTH - 1) / DEPTH;           // batches per block (<= 4)
    const int nfull = DAQP_WG_SCAN_SPLIT ? (c.nblk / c.W) * c.W : c.nblk;
    const int nunits = (c.nblk - nfull) * nb;               // (block, batch) pieces of the left-over blocks
    float *part = reinterpret_cast<float *>(SD(c, red));    // [unit][64]; the reduction area is idle during a scan
    // the image does not depend on u: the first batch of this wave's first block goes out BEFORE u is read
    if (blk0 < nfull) load_batch(blk0, 0);
    float U[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { const int j = lane + 64 * i; U[i] = (j < n) ? (float)SD(c, u)[j] : 0.0f; }
    const double ep = -primal_tol;
    double s1 = DAQP_INF, s2 = DAQP_INF, minq = DAQP_INF, q1 = DAQP_INF, gap1 = 0.0;
    int i1 = kBig, up1 = 0, bad = 0;
    auto ubatch = [&](int b) __attribute__((always_inline)) { return (b == 0) ? U[0] : ((b == 1) ? U[1] : ((b == 2) ? U[2] : U[3])); };
    auto consider = [&](int r, double mu) __attribute__((always_inline)) {
        if (r < c.m) {                                      // (the row's bounds: after the stream, they are not what the pass waits for)
            const int sn = SI(c, sense)[r];
            if (!(sn & (DAQP_ACTIVE + DAQP_IMMUTABLE))) {
                const double du = c.dupper[r], dl = c.dlower[r], sc = c.scaling[r];
                if (!(mu - mu == 0.0)) bad = 1;
                const double cu = du - mu, cl = mu - dl;
                const bool isup = cu <= cl;
                const double s = isup ? cu : cl, q = s - ep * sc, gap = isup ? cl - cu : cu - cl;
                if (q < minq) minq = q;
                if (s < s1) { s2 = s1; s1 = s; i1 = r; up1 = isup ? 1 : 0; q1 = q; gap1 = gap; }
                else if (s < s2) s2 = s;
            }
        }
    };
    for (int blk = blk0; blk < nfull; blk += c.W) {
        float a0 = 0, a1 = 0, a2 = 0, a3 = 0;
        for (int b = 0; b < nb; ++b) {
            if (blk != blk0 || b != 0) load_batch(blk, DEPTH * b);
            const float ub = ubatch(b);
            // (a quad repeated past the end of the row meets u = 0: columns >= n)
#pragma unroll
            for (int q = 0; q < DEPTH; ++q) {
                a0 = __builtin_fmaf(mm[q].x, rlf(ub, 4 * q), a0); a1 = __builtin_fmaf(mm[q].y, rlf(ub, 4 * q + 1), a1);
                a2 = __builtin_fmaf(mm[q].z, rlf(ub, 4 * q + 2), a2); a3 = __builtin_fmaf(mm[q].w, rlf(ub, 4 * q + 3), a3);
            }
        }
        consider(blk * 64 + lane, (double)((a0 + a1) + (a2 + a3)));
    }
    if (nunits > 0) {
        for (int j = wv; j < nunits; j += c.W) {
            const int blk = nfull + j / nb, b = j % nb;
            load_batch(blk, DEPTH * b);
            const float ub = ubatch(b);
            float a0 = 0, a1 = 0, a2 = 0, a3 = 0;
#pragma unroll
            for (int q = 0; q < DEPTH; ++q) {
                a0 = __builtin_fmaf(mm[q].x, rlf(ub, 4 * q), a0); a1 = __builtin_fmaf(mm[q].y, rlf(ub, 4 * q + 1), a1);
                a2 = __builtin_fmaf(mm[q].z, rlf(ub, 4 * q + 2), a2); a3 = __builtin_fmaf(mm[q].w, rlf(ub, 4 * q + 3), a3);
            }
            part[j * 64 + lane] = (a0 + a1) + (a2 + a3);
        }
        __syncthreads();
        for (int l = wv; l < c.nblk - nfull; l += c.W) {    // a left-over block's rows: its batches' sums in batch order
            float mu = part[(l * nb) * 64 + lane];
            for (int b = 1; b < nb; ++b) mu += part[(l * nb + b) * 64 + lane];
            consider((nfull + l) * 64 + lane, (double)mu);
        }
    }
#else
template <int C>
__device__ __forceinline__ void wg_scan32(const WgCtx &c, double primal_tol)
{
    const int wv = wg_wave(), lane = wg_lane(), n = c.n;
    float *u32 = reinterpret_cast<float *>(SD(c, red));     // the reduction area is idle during a scan
    constexpr int DEPTH = DAQP_WG_SCAN_DEPTH;   // 16-byte loads in flight per lane
    const int blk0 = (wv + c.W - 1) % c.W;                  // (wave 0, the master, takes its blocks last)
    float4 mm[DEPTH];
    // the image does not depend on u: the first batch of this wave's first block goes out BEFORE u is converted and published --
    // its trip to memory overlaps the conversion and the barrier instead of following them
    auto load_batch = [&](const float4 *src, int t) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < DEPTH; ++q) { const int tt = (t + q < c.nquad) ? t + q : c.nquad - 1; mm[q] = src[(size_t)tt * 64]; }
    };
    if (blk0 < c.nblk) load_batch(reinterpret_cast<const float4 *>(c.M32) + ((size_t)blk0 * c.nquad) * 64 + lane, 0);
    for (int j = wg_tid(); j < 4 * c.nquad; j += 64 * c.W) u32[j] = (j < n) ? (float)SD(c, u)[j] : 0.0f;
    __syncthreads();
    const double ep = -primal_tol;
    double s1 = DAQP_INF, s2 = DAQP_INF, minq = DAQP_INF, q1 = DAQP_INF, gap1 = 0.0;
    int i1 = kBig, up1 = 0, bad = 0;
    const float4 *u4 = reinterpret_cast<const float4 *>(u32);
    for (int blk = blk0; blk < c.nblk; blk += c.W) {
        const int r = blk * 64 + lane;
        const bool own = r < c.m;
        const int rr = own ? r : 0;
        const float4 *src = reinterpret_cast<const float4 *>(c.M32) + ((size_t)blk * c.nquad) * 64 + lane;
        const double du = c.dupper[rr], dl = c.dlower[rr], sc = c.scaling[rr];
        float a0 = 0, a1 = 0, a2 = 0, a3 = 0;
        for (int t = 0; t < c.nquad; t += DEPTH) {
            if (blk != blk0 || t != 0) load_batch(src, t);
            if (t + DEPTH <= c.nquad) {
#pragma unroll
                for (int q = 0; q < DEPTH; ++q) {
                    const float4 uk = u4[t + q];
                    a0 = __builtin_fmaf(mm[q].x, uk.x, a0); a1 = __builtin_fmaf(mm[q].y, uk.y, a1);
                    a2 = __builtin_fmaf(mm[q].z, uk.z, a2); a3 = __builtin_fmaf(mm[q].w, uk.w, a3);
                }
            } else {
#pragma unroll
                for (int q = 0; q < DEPTH; ++q) {
                    const bool in = t + q < c.nquad;
                    const float4 uk = u4[in ? t + q : 0];
                    const float z = in ? 1.0f : 0.0f;       // (a repeated last load counts zero times)
                    a0 = __builtin_fmaf(mm[q].x * z, uk.x, a0); a1 = __builtin_fmaf(mm[q].y * z, uk.y, a1);
                    a2 = __builtin_fmaf(mm[q].z * z, uk.z, a2); a3 = __builtin_fmaf(mm[q].w * z, uk.w, a3);
                }
            }
        }
        const double mu = (double)((a0 + a1) + (a2 + a3));
        if (own) {
            const int sn = SI(c, sense)[r];
            if (!(sn & (DAQP_ACTIVE + DAQP_IMMUTABLE))) {
                if (!(mu - mu == 0.0)) bad = 1;
                const double cu = du - mu, cl = mu - dl;
                const bool isup = cu <= cl;
                const double s = isup ? cu : cl, q = s - ep * sc, gap = isup ? cl - cu : cu - cl;
                if (q < minq) minq = q;
                if (s < s1) { s2 = s1; s1 = s; i1 = r; up1 = isup ? 1 : 0; q1 = q; gap1 = gap; }
                else if (s < s2) s2 = s;
            }
        }
    }
#endif
    // the wave's best, runner-up and smallest margin
    double bv = s1;
    int bi = i1, aux = lane;
    wave_argmin(bv, bi, aux);                       // aux: the lane that holds the winner
    const double other = (lane == aux && bi != kBig) ? s2 : s1;
    const double w2 = wave_min(other), wq = wave_min(minq);
    const unsigned long long anybad = __ballot(bad);
    const int src_lane = (bi == kBig) ? 0 : aux;
    const double wq1 = rl(q1, src_lane), wgap = rl(gap1, src_lane);
    const int wup = __builtin_amdgcn_readlane(up1, src_lane);
    if (lane == 0) {
        double *o = SD(c, cand) + 8 * wv;
        o[0] = bv; o[1] = w2; o[2] = wq; o[3] = wq1; o[4] = wgap;
        reinterpret_cast<int *>(o + 5)[0] = bi; reinterpret_cast<int *>(o + 5)[1] = wup;
        reinterpret_cast<int *>(o + 6)[0] = anybad ? 1 : 0;
    }
}

// Gram column of the LDL' append (factorization.c:21-60): gram[s] = row(slot s) . m_new for every slot up to `hi`
// (free slots hold stale rows: finite, unused).  lane <-> slot, rows read from the transposed scratch (coalesced).
// Exact mode: the reference's dot_row (four interleaved partial sums from the row's start column, (s0+s1)+(s2+s3));
// otherwise the columns are cut into segments, one per wave, combined in segment order.
template <int C>
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
            const int idk = SI(c, slot_id)[ss];
            const int j0 = (idk < c.ms) ? (c0 > idk ? c0 : idk) : c0;       // factorization.c:64-72
            const int len = n - j0, body = j0 + (len & ~3);
            double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
            for (int jb = 0; jb < n; jb += 8) {
                double rv[8], mv[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) { const int j = (jb + q < n) ? jb + q : n - 1; rv[q] = col[(size_t)j * c.capT]; mv[q] = SD(c, mnew)[j]; }
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
            if (s < hi) SD(c, gram)[s] = (s0 + s1) + (s2 + s3);
        }
        return;
    }
}

// The append's row fetch + Gram column in the default arithmetic, as ONE phase: FOUR slots per wave-load from the row-major scratch --
// 16 lanes x 16 bytes = 256 contiguous bytes of one row each --, a row in ldr / 32 such loads, three groups of four slots in flight
// per lane; every wave works, every lane of it (the lane <-> slot pass over the transposed copy kept 6 of 8 waves and 2/3 of their
// lanes busy with 8-byte loads: 29 k cycles per append for ~200 KB).  The sixteen partial sums of a slot meet in four DPP steps inside
// their row of lanes.  The rows already in the scratch do not depend on the new one: the first batch of their loads is issued
// BEFORE the new row is fetched (one trip to memory for both instead of two in a row); the new row's own entry |m_new|^2 is
// formed from its LDS copy.
template <int C>
__device__ __forceinline__ void wg_fetch_gram_fast(const WgCtx &c, int id, int newslot, int hi)
{
    const int wv = wg_wave(), lane = wg_lane();
    const int l16 = lane & 15, sub = lane >> 4;
    const int nch = c.ldr >> 5;                           // 32-column chunks per row (<= 8: n <= 255)
    const int NG = (hi + 3) >> 2;
    constexpr int GP = 3;
    double2 rv[GP][8];
    auto load_groups = [&](int g0) __attribute__((always_inline)) {
#pragma unroll
        for (int k = 0; k < GP; ++k) {
            const int g = g0 + k * c.W;
            const int sl = 4 * (g < NG ? g : g0) + sub;          // (a group beyond the end re-reads the first: the value is discarded)
            const double2 *row = reinterpret_cast<const double2 *>(c.rowc + (size_t)(sl < c.cap ? sl : 0) * c.ldr) + l16;
#pragma unroll
            for (int i = 0; i < 8; ++i) if (i < nch) rv[k][i] = row[16 * i];
        }
    };
    if (wv < NG) load_groups(wv);
    wg_fetch_row<C>(c, id, newslot, true);
    __syncthreads();
#ifndef DAQP_WG_PROBE2
    if (SI(c, cmd)[6] && wg_tid() == 0) reinterpret_cast<long long *>(wg_sm() + WgL<C>::prof)[22] += (long long)__builtin_readcyclecounter() - SI64(c)[0];   // (probe: the new row is in LDS)
#endif
    double2 mv[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) mv[i] = reinterpret_cast<const double2 *>(SD(c, mnew))[(i < nch ? 16 * i : 0) + l16];
    for (int g0 = wv; g0 < NG; g0 += GP * c.W) {
        if (g0 != wv) load_groups(g0);
#pragma unroll
        for (int k = 0; k < GP; ++k) {
            const int g = g0 + k * c.W, sl = 4 * g + sub;
            const bool self = sl == newslot;                     // (its row in the scratch may not have landed when the load went out)
            double a = 0;
#pragma unroll
            for (int i = 0; i < 8; ++i) if (i < nch) {
                const double rx = self ? mv[i].x : rv[k][i].x, ry = self ? mv[i].y : rv[k][i].y;
                a = __builtin_fma(rx, mv[i].x, a); a = __builtin_fma(ry, mv[i].y, a);
            }
            a += dpp_f64<0xB1>(a);
            a += dpp_f64<0x4E>(a);
            a += dpp_f64<0x141>(a);
            a += dpp_f64<0x140>(a);
            if (g < NG && l16 == 0 && sl < hi) SD(c, gram)[sl] = a;
        }
    }
}

// factorization.c:121-129: move rows r+1.. of packed L up by one and drop column r, element-parallel over the
// destination range [tri(r), tri(na-1)).  A destination always reads from a higher address, so ascending chunks with
// "everybody reads, barrier, everybody writes, barrier" never clobber a live source.
template <int C>
__device__ __forceinline__ void wg_compact(const WgCtx &c, int r, int na)
{
    const int e0 = tri(r), e1 = tri(na - 1);
    const int T = 64 * c.W;
    constexpr int U = 4;
    for (int cb = e0; cb < e1; cb += T * U) {
        double tmp[U];
#pragma unroll
        for (int q = 0; q < U; ++q) {
            const int e = cb + q * T + wg_tid();
            tmp[q] = 0;
            if (e < e1) {
                int i = (int)((sqrtf(8.0f * (float)e + 1.0f) - 1.0f) * 0.5f);
                while (tri(i + 1) <= e) ++i;
                while (tri(i) > e) --i;
                const int j = e - tri(i);
                tmp[q] = SDL(c)[tri(i + 1) + j + (j >= r ? 1 : 0)];
            }
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < U; ++q) {
            const int e = cb + q * T + wg_tid();
            if (e < e1) SDL(c)[e] = tmp[q];
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Inverse factor (default arithmetic mode).  With L in LDS every iteration pays three or four substitution CHAINS on the
// master wave -- na dependent readlane/fma steps each, ~60 k of the 111 k cycles of a C4 iteration, with seven waves
// watching.  Carrying W = L^-1 (unit lower triangular, same packed storage) instead turns each of them into matrix-vector
// products that the whole workgroup shares, with no cross-lane dependency at all:
//   CSP          x = W rhs (only the rows that changed), z = x / D, lam* = W' z
//   append       y = W g, l = y / D, d_new = g_nn - sum y_i l_i, new row of W = -l' W
//   delete row r p = -W[r+1.., r] (the column IS L22^-1 l_r), the rank-one recurrences as prefix sums over
//                t_j = 1/alpha_j (t' = t + p^2/D), then  new rows = K^-1 [W21 + p w_r' | W22]  as one sweep down each column:
//                x = x0 - p_t s,  s += beta_t x  (s: the column's running sum; columns are independent)
// Same mathematics, different rounding (results agree to ~1e-14; the exact mode keeps L and the reference's chains).
// Anything irregular -- a singular or ill-conditioned pivot, pivot_last, refinement, re-activation -- first converts W back
// to L (wg_w2l) and continues on the chains; so does the end of a solve, because the stored iterate is always L.
// ---------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ double wred_sum(double v)
{
    v += dpp_f64<0xB1>(v);
    v += dpp_f64<0x4E>(v);
    v += dpp_f64<0x141>(v);
    v += dpp_f64<0x140>(v);
    return (rl(v, 0) + rl(v, 16)) + (rl(v, 32) + rl(v, 48));
}
// rows from..na-1 of (W v) + v.  SIXTEEN rows per wave and trip, four lanes per row (lane = 4 * row + part: part p sums the columns
// j = p (mod 4), so the four lanes of a row read four adjacent doubles per step); the four partial sums of a row meet in two
// quad-permute steps, and whatever `out` does with a finished row -- a division, a store -- happens for sixteen rows at once.
// (Round 3 summed a row across the whole wave, four rows per trip: 64-lane reductions with read-lanes, ~2.1 k cycles per trip and
// 8.6 k per W g of a 130-row factor -- a single wave issuing ~200 dependent instructions per trip; the LDS traffic itself is 68 KB.)
template <int C, class F>
__device__ __forceinline__ void wg_w_rows(const WgCtx &c, const double *vec, int from, int na, F &&out)
{
    const int wv = wg_wave(), lane = wg_lane();
    const int r = lane >> 2, p = lane & 3;
    for (int i0 = from + 16 * wv; i0 < na; i0 += 16 * c.W) {
        const int i = i0 + r;
        const bool valid = i < na;
        const int irow = valid ? i : na - 1;
        const int ilast = (i0 + 15 < na) ? i0 + 15 : na - 1;
        const int kmax = (ilast + 3) >> 2;                  // steps of four columns that the longest row of the block needs
        const int kfull = i0 >> 2;                          // steps whose columns lie left of EVERY row's diagonal: no test inside
        double a0 = 0, a1 = 0;
        const double *wrow_ = WROW(c, irow);
        const double *wr_ = wrow_ + p, *vp_ = vec + p;
        int k = 0;
        for (; k + 8 <= kfull; k += 8) {
            double wq[8], vq[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) { wq[q] = wr_[4 * (k + q)]; vq[q] = vp_[4 * (k + q)]; }
#pragma unroll
            for (int q = 0; q < 8; q += 2) { a0 = __builtin_fma(wq[q], vq[q], a0); a1 = __builtin_fma(wq[q + 1], vq[q + 1], a1); }
        }
        for (; k < kmax; k += 4) {                          // the rest, column by column against the row's own diagonal
            double wq[4], vq[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int j = 4 * (k + q) + p;
                const bool in = valid && j < i;
                wq[q] = wrow_[in ? j : 0];
                vq[q] = vec[in ? j : 0];
                if (!in) wq[q] = 0.0;
            }
            a0 = __builtin_fma(wq[0], vq[0], a0); a1 = __builtin_fma(wq[1], vq[1], a1);
            a0 = __builtin_fma(wq[2], vq[2], a0); a1 = __builtin_fma(wq[3], vq[3], a1);
        }
        double acc = a0 + a1;
        acc += dpp_f64<0xB1>(acc);    // quad_perm [1,0,3,2]
        acc += dpp_f64<0x4E>(acc);    // quad_perm [2,3,0,1]
        if (p == 0 && valid) out(i, acc + vec[i]);
    }
}
// (W' v)_j = v_j + sum_{i > j} W[i][j] v_i for every column: thread <-> column (adjacent lanes, adjacent addresses), the rows cut
// into slices over the waves that the columns leave idle; partial sums meet in `red`.  All threads; two barriers inside.
template <int C, class F>
__device__ __forceinline__ void wg_w_cols(const WgCtx &c, const double *vec, int na, F &&out)
{
    const int wv = wg_wave(), lane = wg_lane(), tid = wg_tid();
    const int chunks = (na + 63) >> 6;              // <= 3: the inverse factor is only carried while na <= 192
    // waves per 64-column chunk in proportion to the rows below the chunk's first diagonal entry (chunk 0 of a 135-row W has
    // 134 rows to sum, chunk 2 has six): waves [0,b1) serve chunk 0, [b1,b2) chunk 1, [b2,W) chunk 2 (scalars only: a
    // run-time indexed table would live in scratch memory)
    int b1 = c.W, b2 = c.W;
    // (rounded quotients through a float reciprocal: the split only balances work, any split is correct, and every thread computes
    //  the same one; an integer division is ~40 instructions on this hardware and there were three of them per call)
    if (chunks == 2) {
        b1 = (int)((float)(na * c.W) / (float)(2 * na - 64) + 0.5f);
        b1 = b1 < 1 ? 1 : (b1 > c.W - 1 ? c.W - 1 : b1);
    } else if (chunks >= 3) {
        const float inv = 1.0f / (float)(3 * na - 192);
        b1 = (int)((float)(na * c.W) * inv + 0.5f);
        b1 = b1 < 1 ? 1 : (b1 > c.W - 2 ? c.W - 2 : b1);
        b2 = (int)((float)((2 * na - 64) * c.W) * inv + 0.5f);
        b2 = b2 < b1 + 1 ? b1 + 1 : (b2 > c.W - 1 ? c.W - 1 : b2);
    }
    b1 = uni(b1); b2 = uni(b2);
    const int chunk = wv < b1 ? 0 : (wv < b2 ? 1 : 2);
    const int wfirst = chunk == 0 ? 0 : (chunk == 1 ? b1 : b2), wend = chunk == 0 ? b1 : (chunk == 1 ? b2 : c.W);
    const int nsl = wend - wfirst, slice = wv - wfirst;
    double *part = SD(c, red);                      // [wave][64]
    if (chunk < chunks) {
        const int j = chunk * 64 + lane;
        const int rows0 = 64 * chunk + 1, span = na - rows0;          // rows rows0 .. na-1 concern this chunk
        const int per = uni((int)((float)(span + nsl - 1) / (float)(nsl > 0 ? nsl : 1) + 1e-3f));   // (nsl <= 8, span < 256: exact)
        int lo = rows0 + slice * per, hi = lo + per < na ? lo + per : na;
        if (lo < j + 1) lo = j + 1;
        double a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (j < na) {
            int i = lo;
            if (!c.tier) {
                const double *wp = SDL(c) + tri(i) + j;          // W[i][j]; the next row is i + 1 doubles further
                for (; i + 7 < hi; i += 8) {                     // eight rows in flight per trip
                    double wv8[8], vv8[8];
                    const double *q = wp;
#pragma unroll
                    for (int k = 0; k < 8; ++k) { wv8[k] = *q; vv8[k] = vec[i + k]; q += i + k + 1; }
                    wp = q;
#pragma unroll
                    for (int k = 0; k < 8; ++k) a[k] = __builtin_fma(wv8[k], vv8[k], a[k]);
                }
                for (; i < hi; ++i) { a[0] = __builtin_fma(*wp, vec[i], a[0]); wp += i + 1; }
            } else {
                // tiered: the rows below r0 from LDS as above; the rows from r0 on from HBM, sixteen in flight per trip (an L2 round trip each)
                const int hl = hi < c.r0 ? hi : c.r0;
                if (i < hl) {
                    const double *wp = SDL(c) + tri(i) + j;
                    for (; i + 7 < hl; i += 8) {
                        double wv8[8], vv8[8];
                        const double *q = wp;
#pragma unroll
                        for (int k = 0; k < 8; ++k) { wv8[k] = *q; vv8[k] = vec[i + k]; q += i + k + 1; }
                        wp = q;
#pragma unroll
                        for (int k = 0; k < 8; ++k) a[k] = __builtin_fma(wv8[k], vv8[k], a[k]);
                    }
                    for (; i < hl; ++i) { a[0] = __builtin_fma(*wp, vec[i], a[0]); wp += i + 1; }
                }
                if (i < hi) {
                    const DAQP_GLOBAL(double) *gp = as_global(const_cast<const double *>(c.gL)) + tri(i) + j;
                    for (; i + 15 < hi; i += 16) {
                        double wv16[16];
                        const DAQP_GLOBAL(double) *q = gp;
#pragma unroll
                        for (int k = 0; k < 16; ++k) { wv16[k] = *q; q += i + k + 1; }
                        gp = q;
#pragma unroll
                        for (int k = 0; k < 16; ++k) a[k & 7] = __builtin_fma(wv16[k], vec[i + k], a[k & 7]);
                    }
                    for (; i + 3 < hi; i += 4) {
                        double wv4[4];
                        const DAQP_GLOBAL(double) *q = gp;
#pragma unroll
                        for (int k = 0; k < 4; ++k) { wv4[k] = *q; q += i + k + 1; }
                        gp = q;
#pragma unroll
                        for (int k = 0; k < 4; ++k) a[k] = __builtin_fma(wv4[k], vec[i + k], a[k]);
                    }
                    for (; i < hi; ++i) { a[0] = __builtin_fma(*gp, vec[i], a[0]); gp += i + 1; }
                }
            }
        }
        part[wv * 64 + lane] = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
    }
    __syncthreads();
    if (tid < na) {
        const int ck = tid >> 6;
        const int f0 = ck == 0 ? 0 : (ck == 1 ? b1 : b2), f1 = ck == 0 ? b1 : (ck == 1 ? b2 : c.W);
        double sum = vec[tid];
        for (int g = f0; g < f1; ++g) sum += part[g * 64 + (tid & 63)];
        out(tid, sum);
    }
    __syncthreads();
}
template <int C>
__device__ __forceinline__ void wg_wcsp(const WgCtx &c, int from, int na, double *lams)
{
#ifdef DAQP_WG_PROBE2
    WGPROBE_BEGIN(c);
#endif
    wg_w_rows<C>(c, SD(c, rhs), from, na, [&](int i, double xi) __attribute__((always_inline)) {
        SD(c, xl)[i] = xi; SD(c, zl)[i] = xi / SD(c, D)[i];
    });
    __syncthreads();
#ifdef DAQP_WG_PROBE2
    WGPROBE(c, 22);
#endif
    wg_w_cols<C>(c, SD(c, zl), na, [&](int j, double v) __attribute__((always_inline)) { lams[j] = v; });
#ifdef DAQP_WG_PROBE2
    WGPROBE(c, 19);
#endif
}
// after the Gram column: l and the new row of W; leaves each wave's part of sum_i y_i l_i in cand[wave]
template <int C>
__device__ __forceinline__ void wg_wappend(const WgCtx &c, int na)
{
    const int tid = wg_tid();
    double *gp = SD(c, pend_lam), *lv = SD(c, mnew);
    if (tid < na) gp[tid] = SD(c, gram)[SI(c, slot)[tid]];
    __syncthreads();
#ifdef DAQP_WG_PROBE2
    WGPROBE_BEGIN(c);
#endif
    double dpart = 0;                      // (the first lane of every quad carries the rows it finished)
    wg_w_rows<C>(c, gp, 0, na, [&](int i, double yi) __attribute__((always_inline)) {
        const double li = yi / SD(c, D)[i];
        dpart = __builtin_fma(yi, li, dpart);
        lv[i] = li;
    });
    dpart = wred_sum((wg_lane() & 3) == 0 ? dpart : 0.0);
    if (wg_lane() == 0) SD(c, cand)[wg_wave()] = dpart;
    __syncthreads();
#ifdef DAQP_WG_PROBE2
    WGPROBE(c, 20);
#endif
    wg_w_cols<C>(c, lv, na, [&](int j, double v) __attribute__((always_inline)) { WROW(c, na)[j] = -v; });
#ifdef DAQP_WG_PROBE2
    WGPROBE(c, 21);
#endif
}
// delete row / column r (nupd = na - r - 1 >= 1 trailing rows; D[r] and the trailing D checked > 0 by the master)
template <int C>
__device__ __forceinline__ void wg_wdelete(const WgCtx &c, int r, int na, bool xvalid)
{
    const int wv = wg_wave(), lane = wg_lane(), tid = wg_tid(), nupd = na - r - 1;
    double *pvec = SD(c, gram), *bvec = SD(c, pend_lam), *wr = SD(c, mnew);
    double *bnd0 = SD(c, red), *bnd1 = SD(c, red) + 256;      // (two seams: the inverse factor is only carried while na <= 192)
    // phase 0: wave 0 -- p, the recurrences by prefix sums, the new pivots; the others -- row r and the columns at the chunk seams
    if (wv == 0) {
        double carry = 1.0 / SD(c, D)[r];            // t_0 = 1 / alpha_0
#pragma unroll
        for (int cc = 0; cc < C; ++cc) {
            if (64 * cc < nupd) {
                const int t = lane + 64 * cc;
                const bool in = t < nupd;
                double p = 0.0;
                if (in) p = -WROW(c, r + 1 + t)[r];
                const double Dt = in ? SD(c, D)[r + 1 + t] : 1.0;
                double sc = p * p / Dt;                                   // s_t
                // inclusive prefix sum over the wave (Hillis-Steele through ds_bpermute: six steps)
#pragma unroll
                for (int d = 1; d < 64; d <<= 1) { const double up = __shfl_up(sc, d); sc += (lane >= d) ? up : 0.0; }
                const double tnext = carry + sc;                           // t_{t+1}
                const double tprev = carry + __shfl_up(sc, 1) * (lane > 0 ? 1.0 : 0.0);   // t_t
                const double tcur = (lane > 0) ? tprev : carry;
                const double an = 1.0 / tnext, ac = 1.0 / tcur;
                if (in) {
                    pvec[t] = p;
                    bvec[t] = p * an / Dt;                                 // beta_t
                    SD(c, D)[r + t] = __builtin_fma(ac * p, p, Dt);        // dbar_t (position r + t in the new numbering)
                }
                carry = rl(tnext, 63);
            }
        }
    } else {
        for (int cidx = tid - 64; cidx < r; cidx += 64 * (c.W - 1)) wr[cidx] = WROW(c, r)[cidx];
        for (int t = tid - 64; t < nupd; t += 64 * (c.W - 1)) {
            const int ro = r + 1 + t;                                      // old row
            static_for<(C - 1 < 2 ? C - 1 : 2)>([&](auto k) __attribute__((always_inline)) {
                constexpr int cb = 64 * (k + 1);                           // old column read by the last lane of chunk k
                double v = 0.0;
                if (cb < ro) v = WROW(c, ro)[cb];
                if (k == 0) bnd0[t] = v; else bnd1[t] = v;
            });
        }
    }
    __syncthreads();
    // phase 1: one thread per NEW column c' (old column c' or c' + 1), sweeping down the trailing rows; the last thread of
    // the workgroup takes the CSP's x = W rhs along as one more column: x~_2 = K^-1 (x_2 + p x_r) (x_2 + p x_r is what W21 rhs_1
    // + p w_r' rhs_1 + W22 rhs_2 comes to), so that the next CSP has no rows to redo after a removal
    const int cn = tid;
    if (xvalid && tid == 64 * c.W - 1) {
        const double xr = SD(c, xl)[r];
        double s = 0;
        for (int t = 0; t < nupd; ++t) {
            const double pt = pvec[t], bt = bvec[t];
            const double x = __builtin_fma(-pt, s, __builtin_fma(pt, xr, SD(c, xl)[r + 1 + t]));
            s = __builtin_fma(bt, x, s);
            SD(c, xl)[r + t] = x;         // (z = x / D for these rows: by the master, lane-parallel, after the command -- a division
        }                                 //  per step of this ONE thread's serial sweep was what a removal waited for)
    }
    if (cn < na - 1) {
        const bool shift = cn >= r;
        const int co = shift ? cn + 1 : cn;
        const double wrc = shift ? 0.0 : wr[cn];
        const bool seam = shift && (cn & 63) == 63;
        const int k = cn >> 6;
        const double *bnd = (k == 0) ? bnd0 : bnd1;
        // (a column of W22 enters the sweep at its own unit diagonal: x = 1 there, so the running sum starts at beta of that row)
        double s = shift ? bvec[cn - r] : 0.0;
        int t = shift ? cn - r + 1 : 0;                                    // first new row below this column's diagonal: r + t > cn
        if (c.tier) {
            // tiered: eight rows' old entries in flight before their chain (a row beyond r0 is an L2 round trip; left inside the chain, the
            // store of a step and the load of the next one cannot be told apart by the code generator and every step waits for its load).
            // Reading ahead is safe: what this thread reads in rows t .. t+7 is written -- by itself or by its right neighbour -- in later steps.
            for (; t + 7 < nupd; t += 8) {
                double w8[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) w8[k] = seam ? bnd[t + k] : WROW(c, r + 1 + t + k)[co];
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const double pt = pvec[t + k], bt = bvec[t + k];
                    const double x = __builtin_fma(-pt, s, __builtin_fma(pt, wrc, w8[k]));
                    s = __builtin_fma(bt, x, s);
                    WROW(c, r + t + k)[cn] = x;
                }
            }
        }
        for (; t < nupd; ++t) {
            const int ro = r + 1 + t, rn = r + t;
            const double w0 = seam ? bnd[t] : WROW(c, ro)[co];
            const double pt = pvec[t], bt = bvec[t];
            const double x0 = __builtin_fma(pt, wrc, w0);
            const double x = __builtin_fma(-pt, s, x0);
            s = __builtin_fma(bt, x, s);
            WROW(c, rn)[cn] = x;
        }
    }
}
// W -> L in place (rows top to bottom: L[i][j] = -W[i][j] - sum_{j<k<i} W[i][k] L[k][j]), nrows rows
template <int C>
__device__ __forceinline__ void wg_w2l(const WgCtx &c, int nrows)
{
    const int tid = wg_tid();
    double *wrow = SD(c, mnew);
    for (int i = 1; i < nrows; ++i) {
        if (tid < i) wrow[tid] = WROW(c, i)[tid];
        __syncthreads();
        if (tid < i) {
            double acc = -wrow[tid];
            if (!c.tier) {
                for (int k = tid + 1; k < i; ++k) acc = __builtin_fma(-wrow[k], SDL(c)[tri(k) + tid], acc);
            } else {
                int k = tid + 1;
                const int kl = i < c.r0 ? i : c.r0;
                for (; k < kl; ++k) acc = __builtin_fma(-wrow[k], SDL(c)[tri(k) + tid], acc);
                if (k < c.r0) k = c.r0;                       // (tid + 1 may lie beyond r0 already)
                const DAQP_GLOBAL(double) *gp = as_global(const_cast<const double *>(c.gL));
                for (; k + 7 < i; k += 8) {                   // the rows in HBM: eight loads in flight
                    double l8[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) l8[e] = gp[tri(k + e) + tid];
#pragma unroll
                    for (int e = 0; e < 8; ++e) acc = __builtin_fma(-wrow[k + e], l8[e], acc);
                }
                for (; k < i; ++k) acc = __builtin_fma(-wrow[k], gp[tri(k) + tid], acc);
            }
            WROW(c, i)[tid] = acc;
        }
        __syncthreads();
    }
}

// one command, executed by every wave (the master included)
template <int C>
__device__ __forceinline__ void wg_do(const WgCtx &c, int code, double primal_tol)
{
    const int a0 = uni(SI(c, cmd)[1]), a1 = uni(SI(c, cmd)[2]), na = uni(SI(c, cmd)[3]), hi = uni(SI(c, cmd)[4]);
    const double *lams = uni(SI(c, cmd)[5]) ? SD(c, lamB) : SD(c, lamA);
    if (code == WG_PRIMAL) wg_primal<C>(c, na, lams);
    else if (code == WG_SCAN) wg_scan<C>(c, primal_tol);
    else if (code == WG_SCAN32) wg_scan32<C>(c, primal_tol);
    else if (code == WG_FETCH_GRAM) {
        if (c.exact) {
            wg_fetch_row<C>(c, a0, a1, true);
            __syncthreads();
            wg_gram<C>(c, a0, hi);
        } else wg_fetch_gram_fast<C>(c, a0, a1, hi);
    } else if (code == WG_COMPACT) wg_compact<C>(c, a0, na);
    else if (code == WG_WCSP) wg_wcsp<C>(c, a0, na, const_cast<double *>(lams));
    else if (code == WG_WAPPEND) {          // (the inverse factor exists in the default arithmetic only)
        const bool pr = SI(c, cmd)[6] != 0 && wg_tid() == 0;
        if (pr) SI64(c)[0] = (long long)__builtin_readcyclecounter();
        wg_fetch_gram_fast<C>(c, a0, a1, hi);
        __syncthreads();
        const long long t1 = pr ? (long long)__builtin_readcyclecounter() : 0;
        wg_wappend<C>(c, na);
#ifndef DAQP_WG_PROBE2
        if (pr) { long long *pp = reinterpret_cast<long long *>(wg_sm() + WgL<C>::prof); pp[20] += t1 - SI64(c)[0]; pp[21] += (long long)__builtin_readcyclecounter() - t1; }
#endif
    } else if (code == WG_WDELETE) wg_wdelete<C>(c, a0, na, a1 != 0);
    else if (code == WG_W2L) wg_w2l<C>(c, a0);
}

// master side: post a command, take part in it
template <int C>
__device__ __forceinline__ void wg_run(WgWave<C> &w, int code, int a0 = 0, int a1 = 0)
{
    const WgCtx &c = w.c;
    if (wg_lane() == 0) {
        SI(c, cmd)[0] = code; SI(c, cmd)[1] = a0; SI(c, cmd)[2] = a1; SI(c, cmd)[3] = w.na; SI(c, cmd)[4] = w.hi_slot + 1;
        SI(c, cmd)[5] = w.lam_b ? 0 : 1;   // which buffer holds lam*
        SI(c, cmd)[6] = w.profiling ? 1 : 0;
    }
    __syncthreads();
    const long long tA = w.profiling ? (long long)__builtin_readcyclecounter() : 0;
    wg_do<C>(c, code, w.stp->primal_tol);
    const long long tB = w.profiling ? (long long)__builtin_readcyclecounter() : 0;
    __syncthreads();
    if (w.profiling && (code == WG_SCAN32) && wg_lane() == 0) {
        const long long tC = (long long)__builtin_readcyclecounter();
        reinterpret_cast<long long *>(wg_sm() + WgL<C>::prof)[18] += tB - tA;
#ifndef DAQP_WG_PROBE2
        reinterpret_cast<long long *>(wg_sm() + WgL<C>::prof)[19] += tC - tB;
#endif
    }
}
// everybody else: serve commands until the master says EXIT
template <int C>
__device__ __forceinline__ void wg_serve(const WgCtx &c, double primal_tol)
{
    for (;;) {
        __syncthreads();
        const int code = uni(SI(c, cmd)[0]);
        if (code == WG_EXIT) break;
        wg_do<C>(c, code, primal_tol);
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// master: the serial part of the iteration on wave 0 (lane + 64 c <-> working-set position), L and the vectors in LDS
// ---------------------------------------------------------------------------------------------------------------------
constexpr int kWPre = 8;

// Cross-lane reads of a vector held as a[cc] in lane (idx & 63), chunk (idx >> 6).  A register array must never be indexed
// with a run-time value -- "v = a[0]; if (chunk == 1) v = a[1]; ..." is folded into exactly that, the array moves to scratch
// memory and every access of the chains below becomes a trip to it (measured: 600 cycles per substitution step).  Here
// the chunk is chosen by a wave-uniform BRANCH around code whose array index is a compile-time constant.
template <int C>
__device__ __forceinline__ double wpick(const double (&a)[C], int idx)
{
    double v = 0;
    static_for<C>([&](auto cc) __attribute__((always_inline)) { if ((idx >> 6) == cc) v = rl(a[cc], idx & 63); });
    return v;
}
// f(cs, q, x) for the eight pivots 8*g8 + q, q = 0..7 (or 7..0 when `down`), x = the CURRENT value of that element of a
// (f may update a between pivots); cs = the pivots' chunk as a compile-time constant
template <int C, bool DOWN, class F>
__device__ __forceinline__ void wpivots8(double (&a)[C], int g8, F &&f)
{
    const int l0 = (g8 & 7) * 8;
    static_for<C>([&](auto cs) __attribute__((always_inline)) {
        if ((g8 >> 3) == cs) {
            static_for<8>([&](auto qq) __attribute__((always_inline)) {
                constexpr int q = DOWN ? 7 - (int)qq : (int)qq;
                f(cs, std::integral_constant<int, q>{}, rl(a[cs], l0 + q));
            });
        }
    });
}

// A chain is a run of such groups; a taken branch costs as much as the arithmetic of a pivot (tools/ubench.hip: ~40
// cycles), so the per-pivot end-of-range test exists only in the one partial group of a chain: body(cs, q, x, guard) with
// guard a compile-time bool.
template <int C, bool DOWN, class F>
__device__ __forceinline__ void wgroup(double (&a)[C], int g8, bool full, F &&body)
{
    if (full) wpivots8<C, DOWN>(a, g8, [&](auto cs, auto q, double x) __attribute__((always_inline)) { body(cs, q, x, std::false_type{}); });
    else wpivots8<C, DOWN>(a, g8, [&](auto cs, auto q, double x) __attribute__((always_inline)) { body(cs, q, x, std::true_type{}); });
}
// sum over the whole wave of a chunked vector (default arithmetic mode only: not the reference's order)
template <int C>
__device__ __forceinline__ double wsum(const double (&a)[C])
{
    double v = a[0];
    static_for<C - 1>([&](auto cc) __attribute__((always_inline)) { v += a[cc + 1]; });
    v += dpp_f64<0xB1>(v);    // quad_perm [1,0,3,2]
    v += dpp_f64<0x4E>(v);    // quad_perm [2,3,0,1]
    v += dpp_f64<0x141>(v);   // row_half_mirror
    v += dpp_f64<0x140>(v);   // row_mirror
    return (rl(v, 0) + rl(v, 16)) + (rl(v, 32) + rl(v, 48));
}

// leave the inverse-factor representation: W -> L over the first `rows` rows, then the chains take over for the rest of the solve
template <int C>
__device__ __forceinline__ void wleave_w(WgWave<C> &w, int rows, bool final = false)
{
    if (!w.use_w) return;
    // a tiered factor cannot continue on the chains (they walk L in LDS): the problem is handed to the launch behind this one, which holds
    // the whole factor in LDS; only the END of a solve converts across the tiers (final: the stored iterate is always L)
    if (w.c.tier && rows > w.c.r0 && !final) { w.overflow = 1; return; }
    wg_run(w, WG_W2L, rows);
    w.use_w = 0;
}

// LDL' row append (factorization.c:21-111)
template <int C>
__device__ __forceinline__ void wldl_append(WgWave<C> &w, int id)
{
    if (w.use_w && w.na >= 191) wleave_w(w, w.na);   // (the inverse-factor commands serve up to three 64-row chunks: 192 rows)
    if (w.overflow) return;
    if (w.use_w) {   // inverse factor: Gram column, l = D^-1 W g, new row of W = -l' W, all in one command
        const WgCtx &c = w.c;
        const int lane = wg_lane(), na = w.na;
        if (na >= c.capW) { w.overflow = 1; return; }
        const int newslot = uni(SI(c, freestk)[w.nfree - 1]);
        w.nfree--;
        if (newslot > w.hi_slot) w.hi_slot = newslot;
        if (lane == 0) {
            SI(c, slot)[na] = newslot; SI(c, slot_id)[newslot] = id;
            SD(c, rhs)[na] = (SI(c, sense)[id] & DAQP_LOWER) ? -c.dlower[id] : -c.dupper[id];
        }
        WPROF_T0(w);
        wg_run(w, WG_WAPPEND, id, newslot);
        WPROF_ACC(w, 8);
        w.sing = kEmpty;
        double dsum = 0;
        for (int k = 0; k < c.W; ++k) dsum += SD(c, cand)[k];
        const double dnew = und(SD(c, gram)[newslot] - dsum);
        if (ub(dnew < w.stp->sing_tol) || na >= c.n) {
            // a singular pivot: back to L (the new row included: its L entries are l), then as the chains would leave it
            wleave_w(w, na + 1);
            if (w.overflow) return;
            if (lane == 0) SD(c, D)[na] = 0;
            w.sing = na;
        } else { if (lane == 0) SD(c, D)[na] = dnew; w.fast_na = na + 1; }
        WSYNC();
        WPROF_ACC(w, 9);
        return;
    }
    const WgCtx &c = w.c;
    const int lane = wg_lane(), na = w.na, n = c.n, base = tri(na);
    if (na >= c.capL) { w.overflow = 1; return; }          // packed L would outgrow its LDS: the one-wave kernel takes this problem
    // a free slot of the active-row scratch (lowest first: the slots in use stay dense)
    const int newslot = uni(SI(c, freestk)[w.nfree - 1]);
    w.nfree--;
    if (newslot > w.hi_slot) w.hi_slot = newslot;
    if (lane == 0) { SI(c, slot)[na] = newslot; SI(c, slot_id)[newslot] = id; }
    WPROF_T0(w);
    wg_run(w, WG_FETCH_GRAM, id, newslot);
    WPROF_ACC(w, 8);
    w.sing = kEmpty;
    double g[C];
    int ns_act = 0;
#pragma unroll
    for (int cc = 0; cc < C; ++cc) {
        const int k = lane + 64 * cc;
        g[cc] = 0;
        int soft_k = 0;
        if (k <= na) {
            g[cc] = SD(c, gram)[SI(c, slot)[k]];
            const int idk = (k < na) ? SI(c, ws)[k] : id;
            soft_k = (SI(c, sense)[idk] & DAQP_SOFT) ? 1 : 0;
        }
        if (w.has_soft) ns_act += __popcll(__ballot(soft_k));
    }
    double dnew = wpick<C>(g, na);
    if (ub((SI(c, sense)[id] & DAQP_SOFT) != 0)) dnew += w.stp->rho_soft;
    if (na == 0) {
        if (lane == 0) SD(c, D)[0] = dnew;
        WSYNC();
        WPROF_ACC(w, 9);
        return;
    }
    // forward substitution  l <- L \ g   (factorization.c:81-88): pivots j ascending, eight at a time, the lane's own L entries
    // of the group loaded before its chain
    // (Rows of chunks above the pivots' own chunk take every update of the group unconditionally -- lanes beyond the working
    //  set compute on whatever L holds there and are never read; only the pivots' chunk needs the k > j select.)
    for (int g8 = 0; 8 * g8 < na - 1; ++g8) {
        double Lv[kWPre][C];
        static_for<C>([&](auto cc) __attribute__((always_inline)) {   // (one uniform branch per chunk, not per load: a branch costs as much as four loads)
            if (64 * (cc + 1) > 8 * g8 && 64 * cc < na) {
                const int rowbase = tri(lane + 64 * cc) + 8 * g8;
                static_for<kWPre>([&](auto q) __attribute__((always_inline)) { Lv[q][cc] = SDL(c)[WLIDX(c, rowbase + q)]; });
            }
        });
        wgroup<C, false>(g, g8, 8 * g8 + 8 <= na - 1, [&](auto cs, auto q, double lj, auto guard) __attribute__((always_inline)) {
            const int j = 8 * g8 + q;
            if constexpr (guard) { if (j >= na - 1) return; }
            static_for<C>([&](auto ct) __attribute__((always_inline)) {
                if constexpr (ct == cs) {
                    const double t = g[ct] - Lv[q][ct] * lj;
                    g[ct] = (lane + 64 * ct > j) ? t : g[ct];
                } else if constexpr (ct > cs) {
                    if (64 * ct < na) g[ct] = g[ct] - Lv[q][ct] * lj;
                }
            });
        });
    }
    // l_k /= D_k ; d_new -= sum_k l_k^2 D_k, in k order (factorization.c:93-103)
    double p[C];
#pragma unroll
    for (int cc = 0; cc < C; ++cc) {
        const int k = lane + 64 * cc;
        p[cc] = 0;
        if (k < na) {
            const double t = g[cc];
            const double lk = t / SD(c, D)[k];
            SDL(c)[base + k] = lk;
            p[cc] = t * lk;
        }
    }
    double acc = dnew;
    if (c.exact) {   // the reference's order (p is exactly +0.0 beyond the working set: no end-of-range test)
        for (int g8 = 0; 8 * g8 < na; ++g8)
            wpivots8<C, false>(p, g8, [&](auto cs, auto q, double pk) __attribute__((always_inline)) { acc -= pk; });
    } else acc -= wsum<C>(p);
    int sing = kEmpty;
    if (ub(acc < w.stp->sing_tol) || na >= n + ns_act) { sing = na; acc = 0; }
    if (lane == 0) SD(c, D)[na] = acc;
    w.sing = sing;
    WSYNC();
    WPROF_ACC(w, 9);
}

// LDL' row delete (factorization.c:112-151)
template <int C>
__device__ __forceinline__ void wldl_delete(WgWave<C> &w, int r)
{
    const WgCtx &c = w.c;
    const int lane = wg_lane(), na = w.na;
    if (na == r + 1) return;
    const int nupd = na - r - 1;
    if (w.use_w) {
        // the prefix-sum form of the recurrences needs D[r] > 0 and a positive trailing block
        double dmin = SD(c, D)[r];
#pragma unroll
        for (int cc = 0; cc < C; ++cc) {
            const int t = lane + 64 * cc;
            if (t < nupd) { const double dt = SD(c, D)[r + 1 + t]; dmin = dt < dmin ? dt : dmin; }
        }
        if (ub(wave_min(dmin) > 1e-200)) {
            const int xvalid = (w.reuse >= na) ? 1 : 0;     // x = W rhs complete (the removal follows a CSP): updated by the sweep
            WPROF_T0(w);
            wg_run(w, WG_WDELETE, r, xvalid);
            if (xvalid) {
#pragma unroll
                for (int cc = 0; cc < C; ++cc) {            // z = x / D over the rows whose x the sweep carried (new numbering r .. na-2)
                    const int i = lane + 64 * cc;
                    if (i >= r && i < na - 1) SD(c, zl)[i] = SD(c, xl)[i] / SD(c, D)[i];
                }
                WSYNC();
            }
            WPROF_ACC(w, 11);
            if (xvalid) w.reuse = na;                       // (wdrop_core lowers it to na - 1 = all rows of the new factor)
            return;
        }
        wleave_w(w, na);
        if (w.overflow) return;
    }
    double wv[C];
#pragma unroll
    for (int cc = 0; cc < C; ++cc) {
        const int t = lane + 64 * cc;
        wv[cc] = (t < nupd) ? SDL(c)[tri(r + 1 + t) + r] : 0.0;
    }
    WSYNC();
    WPROF_T0(w);
    wg_run(w, WG_COMPACT, r);
    WPROF_ACC(w, 10);
    double alpha = SD(c, D)[r];
    for (int g8 = 0; 8 * g8 < nupd; ++g8) {
        // the lane's own L entries of the group's eight columns (new numbering: row r+t, column r+j) before the chain, written back after
        double Lc[kWPre][C], Dv[kWPre];
        static_for<kWPre>([&](auto q) __attribute__((always_inline)) {
            const int j = 8 * g8 + q;
            Dv[q] = SD(c, D)[(j < nupd) ? r + 1 + j : r];
        });
        static_for<C>([&](auto cc) __attribute__((always_inline)) {
            if (64 * (cc + 1) > 8 * g8 && 64 * cc < nupd) {
                const int rowbase = tri(r + lane + 64 * cc) + r + 8 * g8;
                static_for<kWPre>([&](auto q) __attribute__((always_inline)) { Lc[q][cc] = SDL(c)[WLIDX(c, rowbase + q)]; });
            }
        });
        wgroup<C, false>(wv, g8, 8 * g8 + 8 <= nupd, [&](auto cs, auto q, double p, auto guard) __attribute__((always_inline)) {
            const int j = 8 * g8 + q;
            if constexpr (guard) { if (j >= nupd) return; }
            {
                const double Di = Dv[q];
                const double dbar = Di + alpha * p * p;
                const double beta = p * alpha / dbar;
                alpha = Di * alpha / dbar;
                if (lane == 0) SD(c, D)[r + j] = dbar;
                static_for<C>([&](auto ct) __attribute__((always_inline)) {
                    if constexpr (ct == cs) {
                        const bool on = lane + 64 * ct > j;
                        const double wn = wv[ct] - p * Lc[q][ct];
                        wv[ct] = on ? wn : wv[ct];
                        const double ln = Lc[q][ct] + beta * wv[ct];
                        Lc[q][ct] = on ? ln : Lc[q][ct];
                    } else if constexpr (ct > cs) {
                        if (64 * ct < nupd) {
                            wv[ct] = wv[ct] - p * Lc[q][ct];
                            Lc[q][ct] = Lc[q][ct] + beta * wv[ct];
                        }
                    }
                });
            }
        });
        static_for<C>([&](auto cc) __attribute__((always_inline)) {
            if (64 * (cc + 1) > 8 * g8 && 64 * cc < nupd) {
                const int t = lane + 64 * cc, rowbase = tri(r + t) + r + 8 * g8;
                static_for<kWPre>([&](auto q) __attribute__((always_inline)) {
                    const int j = 8 * g8 + q;
                    if (j < nupd && t > j && t < nupd) SDL(c)[rowbase + q] = Lc[q][cc];
                });
            }
        });
    }
    WSYNC();
    WPROF_ACC(w, 11);
}

// auxiliary.c:3-22 without the trailing pivot; returns 1 if the factor became singular
template <int C>
__device__ __forceinline__ int wdrop_core(WgWave<C> &w, int r)
{
    const WgCtx &c = w.c;
    const int lane = wg_lane();
    const int idr = uni(SI(c, ws)[r]);
    wtrace(w, -(idr + 1));
    if (lane == 0) { SI(c, sense)[idr] &= ~DAQP_ACTIVE; SI(c, freestk)[w.nfree] = SI(c, slot)[r]; }
    w.nfree++;
    wldl_delete(w, r);
    w.na--;
    w.fast_na = -1;
    int wsn[C], sln[C];
    double lmn[C], rhn[C];
#pragma unroll
    for (int cc = 0; cc < C; ++cc) {
        const int i = lane + 64 * cc;
        wsn[cc] = 0; sln[cc] = 0; lmn[cc] = 0; rhn[cc] = 0;
        if (i >= r && i < w.na) { wsn[cc] = SI(c, ws)[i + 1]; sln[cc] = SI(c, slot)[i + 1]; lmn[cc] = WLAM(w)[i + 1]; rhn[cc] = SD(c, rhs)[i + 1]; }
    }
    WSYNC();
#pragma unroll
    for (int cc = 0; cc < C; ++cc) {
        const int i = lane + 64 * cc;
        if (i >= r && i < w.na) { SI(c, ws)[i] = wsn[cc]; SI(c, slot)[i] = sln[cc]; WLAM(w)[i] = lmn[cc]; SD(c, rhs)[i] = rhn[cc]; }
    }
    if (w.use_w && w.reuse > w.na) w.reuse = w.na;        // inverse factor: x was carried through the removal
    else if (r < w.reuse) w.reuse = r;
    int took = 0;
    WSYNC();
    if (w.na > 0 && ub(SD(c, D)[w.na - 1] < w.stp->sing_tol)) {
        wleave_w(w, w.na);
        w.sing = w.na - 1;
        took = w.overflow ? 0 : 1;
    }
    WSYNC();
    if (took && lane == 0) SD(c, D)[w.na - 1] = 0;
    WSYNC();
    return took;
}

template <int C>
__device__ __forceinline__ void wpush_core(WgWave<C> &w, int id, double lamv) // auxiliary.c:27-40
{
    const WgCtx &c = w.c;
    const int lane = wg_lane();
    wtrace(w, id + 1);
    if (lane == 0) SI(c, sense)[id] |= DAQP_ACTIVE;
    WSYNC();
    wldl_append(w, id);
    if (w.overflow) return;
    if (lane == 0) { SI(c, ws)[w.na] = id; WLAM(w)[w.na] = lamv; }
    w.na++;
    WSYNC();
}

// b <- L' \ b for the leading cnt rows (product order b_j * L[j][i]): pivots j = cnt-1 ... 1, eight at a time
template <int C>
__device__ __forceinline__ void wbackward(WgWave<C> &w, double (&b)[C], int cnt)
{
    const WgCtx &c = w.c;
    const int lane = wg_lane();
    for (int g8 = (cnt - 1) >> 3; g8 >= 0; --g8) {
        double Lv[kWPre][C];
        WPROF_T0(w);
        static_for<C>([&](auto cc) __attribute__((always_inline)) {
            if (64 * cc < 8 * g8 + 8)   // row j of L, lane <-> column (entries beyond the diagonal are loaded and never used)
                static_for<kWPre>([&](auto q) __attribute__((always_inline)) { Lv[q][cc] = SDL(c)[WLIDX(c, tri(8 * g8 + q) + lane + 64 * cc)]; });
        });
        if (w.profiling) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
        WPROF_ACC(w, 13);
        wgroup<C, true>(b, g8, g8 >= 1 && 8 * g8 + 8 <= cnt, [&](auto cs, auto q, double bj, auto guard) __attribute__((always_inline)) {
            const int j = 8 * g8 + q;
            if constexpr (guard) { if (j < 1 || j >= cnt) return; }
            static_for<C>([&](auto ct) __attribute__((always_inline)) {
                if constexpr (ct == cs) {
                    const double t = b[ct] - bj * Lv[q][ct];
                    b[ct] = (lane + 64 * ct < j) ? t : b[ct];
                } else if constexpr (ct < cs) b[ct] = b[ct] - bj * Lv[q][ct];   // whole chunks below the pivot
            });
        });
        WPROF_ACC(w, 14);
    }
}
// x_i = rhs_i - sum_{j<i} L[i][j] x_j for rows i >= from (j ascending); rows < from are final in xl
template <int C>
__device__ __forceinline__ void wforward(WgWave<C> &w, double (&acc)[C], int from)
{
    const WgCtx &c = w.c;
    const int lane = wg_lane(), na = w.na;
    if (from == na - 1 && na > 1) {
        // the usual case after an add: only the last row is open -- its products in parallel, then the j-ordered chain of
        // subtractions (the same operations, in the same order, as the sweep below performs for that row)
        double p[C];
#pragma unroll
        for (int cc = 0; cc < C; ++cc) {
            const int j = lane + 64 * cc;
            p[cc] = (j < na - 1) ? SDL(c)[tri(na - 1) + j] * SD(c, xl)[j] : 0.0;
        }
        double last = wpick<C>(acc, na - 1);
        if (c.exact) {   // (p is exactly +0.0 from column na-1 on: no end-of-range test)
            for (int g8 = 0; 8 * g8 < na - 1; ++g8)
                wpivots8<C, false>(p, g8, [&](auto cs, auto q, double pj) __attribute__((always_inline)) { last -= pj; });
        } else last -= wsum<C>(p);
        static_for<C>([&](auto cc) __attribute__((always_inline)) { acc[cc] = (lane + 64 * cc == na - 1) ? last : acc[cc]; });
        return;
    }
    if (w.profiling && lane == 0) reinterpret_cast<long long *>(wg_sm() + WgL<C>::prof)[15] += 1;   // general sweeps (after a removal)
    // rows < from are final: the chunks that hold only such rows are skipped, the chunk that holds row `from` selects
    const int cfrom = from >> 6;
    // Pivots j < from are final values: their contribution to the open rows is a plain (row . x) accumulation with no
    // cross-lane dependency -- every lane walks its own row of L, x_j comes as an LDS broadcast, j ascending (the reference's
    // order for each row).  Rows < from compute on whatever they hold and are never read.  Only the pivots from `from` on
    // form a triangular solve.
    int g8 = 0;
    for (; 8 * g8 + 8 <= from; ++g8) {
        double Lv[kWPre][C], xf[kWPre];
        static_for<kWPre>([&](auto q) __attribute__((always_inline)) { xf[q] = SD(c, xl)[8 * g8 + q]; });
        static_for<C>([&](auto cc) __attribute__((always_inline)) {
            if (64 * cc < na && cc >= cfrom) {
                const int rowbase = tri(lane + 64 * cc) + 8 * g8;
                static_for<kWPre>([&](auto q) __attribute__((always_inline)) { Lv[q][cc] = SDL(c)[WLIDX(c, rowbase + q)]; });
            }
        });
        static_for<C>([&](auto cc) __attribute__((always_inline)) {
            if (64 * cc < na && cc >= cfrom)
                static_for<kWPre>([&](auto q) __attribute__((always_inline)) { acc[cc] = acc[cc] - Lv[q][cc] * xf[q]; });
        });
    }
    for (; 8 * g8 < na - 1; ++g8) {
        double Lv[kWPre][C], xf[kWPre];
        static_for<kWPre>([&](auto q) __attribute__((always_inline)) {
            const int j = 8 * g8 + q;
            xf[q] = SD(c, xl)[(j < from) ? j : 0];
        });
        static_for<C>([&](auto cc) __attribute__((always_inline)) {
            if (64 * (cc + 1) > 8 * g8 && 64 * cc < na && cc >= cfrom) {
                const int rowbase = tri(lane + 64 * cc) + 8 * g8;
                static_for<kWPre>([&](auto q) __attribute__((always_inline)) { Lv[q][cc] = SDL(c)[WLIDX(c, rowbase + q)]; });
            }
        });
        wgroup<C, false>(acc, g8, 8 * g8 + 8 <= na - 1, [&](auto cs, auto q, double aj, auto guard) __attribute__((always_inline)) {
            const int j = 8 * g8 + q;
            if constexpr (guard) { if (j >= na - 1) return; }
            {
                const double xj = (j < from) ? xf[q] : aj;
                static_for<C>([&](auto ct) __attribute__((always_inline)) {
                    if constexpr (ct >= cs) {
                        if (64 * ct < na && ct >= cfrom) {
                            const int i = lane + 64 * ct;
                            const double t = acc[ct] - Lv[q][ct] * xj;
                            if (ct == cs || ct == cfrom) acc[ct] = (i >= from && i > j) ? t : acc[ct];
                            else acc[ct] = t;
                        }
                    }
                });
            }
        });
    }
}

// lam* of the iteration: the constrained stationary point L D L' lam* = -d (auxiliary.c:314-354) or, with a singular factor,
// the singular direction (auxiliary.c:357-376).  Both end in the one backward substitution of this kernel.
template <int C>
__device__ __forceinline__ void wdirection(WgWave<C> &w)
{
    const WgCtx &c = w.c;
    const int lane = wg_lane(), na = w.na;
    const bool regular = (w.sing == kEmpty);
    if (w.use_w) {    // (regular by construction: a singular pivot leaves the inverse-factor representation at once)
        WPROF_T0(w);
        if (w.fast_na == na && w.reuse == na - 1) {
            // Right after a regular append nothing about the old rows has changed: x_i = (W rhs)_i and z_i = x_i / D_i stand, the
            // new row of W is [-l' W, 1], so x_new = rhs_new - l . x_old, and with lam*_old = W_old' z_old still in the buffer that
            // the add's pointer swap (auxiliary.c:159-160) turned into lam:  lam*_j = lam_j + z_new W[new][j],  lam*_new = z_new.
            // O(na) on this wave, no pass over W and no workgroup command (the full CSP: ~2 k + ~3.9 k cycles and four barriers).
            const int nw = na - 1;
            const double *lv = SD(c, mnew), *lam = WLAM(w);
            double *lams = WLAMS(w);
            double part[C], lold[C], wn[C];
#pragma unroll
            for (int cc = 0; cc < C; ++cc) {
                const int i = lane + 64 * cc;
                const bool in = i < nw;
                part[cc] = in ? lv[i] * SD(c, xl)[i] : 0.0;
                lold[cc] = in ? lam[i] : 0.0;
                wn[cc] = 0.0;
                if (in) wn[cc] = WROW(c, nw)[i];
            }
            const double xn = und(SD(c, rhs)[nw] - wsum<C>(part));
            const double zn = und(xn / SD(c, D)[nw]);
#pragma unroll
            for (int cc = 0; cc < C; ++cc) {
                const int i = lane + 64 * cc;
                if (i < nw) lams[i] = __builtin_fma(zn, wn[cc], lold[cc]);
            }
            if (lane == 0) { SD(c, xl)[nw] = xn; SD(c, zl)[nw] = zn; lams[nw] = zn; }
            WSYNC();
        } else wg_run(w, WG_WCSP, w.reuse);
        w.fast_na = -1;
        WPROF_ACC(w, 7);
        w.reuse = na;
        return;
    }
    double b[C];
    int cnt;
    if (regular) {
        const int from = w.reuse;
        double acc[C];
#pragma unroll
        for (int cc = 0; cc < C; ++cc) {
            const int i = lane + 64 * cc;
            acc[cc] = 0;
            if (i >= from && i < na) {
                const int id = SI(c, ws)[i];
                acc[cc] = (SI(c, sense)[id] & DAQP_LOWER) ? -c.dlower[id] : -c.dupper[id];
            }
        }
        WPROF_T0(w);
        wforward<C>(w, acc, from);
        WPROF_ACC(w, 6);
#pragma unroll
        for (int cc = 0; cc < C; ++cc) {
            const int i = lane + 64 * cc;
            b[cc] = 0;
            if (i < na) {
                if (i >= from) {
                    SD(c, xl)[i] = acc[cc];
                    b[cc] = acc[cc] / SD(c, D)[i];
                    SD(c, zl)[i] = b[cc];
                } else b[cc] = SD(c, zl)[i];
            }
        }
        cnt = na;
    } else {
        const int s_ = w.sing, base = tri(s_);
#pragma unroll
        for (int cc = 0; cc < C; ++cc) {
            const int i = lane + 64 * cc;
            b[cc] = (i < s_) ? -SDL(c)[base + i] : 0.0;
        }
        cnt = s_;
    }
    {
        WPROF_T0(w);
        wbackward<C>(w, b, cnt);
        WPROF_ACC(w, 7);
    }
    double *lams = WLAMS(w);
    if (regular) {
#pragma unroll
        for (int cc = 0; cc < C; ++cc) {
            const int i = lane + 64 * cc;
            if (i < na) lams[i] = b[cc];
        }
        w.reuse = na;
    } else {
        const int s_ = w.sing;
        const bool flip = (SI(c, sense)[SI(c, ws)[s_]] & DAQP_LOWER) != 0;
#pragma unroll
        for (int cc = 0; cc < C; ++cc) {
            const int i = lane + 64 * cc;
            if (i <= s_) {
                const double v = (i == s_) ? 1.0 : b[cc];
                lams[i] = flip ? -v : v;
            }
        }
    }
    WSYNC();
}

// ratio test of auxiliary.c:277-311 (SOFT_WEIGHTS off): the position to drop (or kBig) after stepping lam towards lam*;
// the removal itself is the state machine's single DROP site
template <int C>
__device__ __forceinline__ int wblocking_test(WgWave<C> &w)
{
    const WgCtx &c = w.c;
    const int lane = wg_lane(), na = w.na;
    const double dtol = w.stp->dual_tol;
    const bool regular = (w.sing == kEmpty);
    double bv = DAQP_INF;
    int bi = kBig, aux = 0;
    double lm[C], ls[C];
    double *lam = WLAM(w);
    const double *lams = WLAMS(w);
#pragma unroll
    for (int cc = 0; cc < C; ++cc) {
        const int i = lane + 64 * cc;
        lm[cc] = 0; ls[cc] = 0;
        if (i < na) {
            lm[cc] = lam[i]; ls[cc] = lams[i];
            const int sn = SI(c, sense)[SI(c, ws)[i]];
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
    if (bi == kBig) return kBig;
    const double alpha = bv;
#pragma unroll
    for (int cc = 0; cc < C; ++cc) {
        const int i = lane + 64 * cc;
        if (i < na) lam[i] = regular ? lm[cc] + alpha * (ls[cc] - lm[cc]) : lm[cc] + alpha * ls[cc];
    }
    w.sing = kEmpty;
    WSYNC();
    return bi;
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
            if (ub((SI(c, sense)[SI(c, ws)[i]] & DAQP_SOFT) != 0)) { const double li = WLAMS(w)[i]; fv += li * li; }
    }
    fv = fv * w.stp->rho_soft;
    w.soft = und(fv);
}
// start + u_0^2 + u_1^2 + ... in index order (auxiliary.c:85-86): lane <-> component for the squares, then the ordered chain
// over broadcast operands
template <int C>
__device__ __forceinline__ double wordered_norm2(WgWave<C> &w, double start)
{
    const WgCtx &c = w.c;
    const int lane = wg_lane();
    double sq[4];
    static_for<4>([&](auto cc) __attribute__((always_inline)) {
        const int j = lane + 64 * cc;
        const double uj = (j < c.n) ? SD(c, u)[j] : 0.0;
        sq[cc] = uj * uj;
    });
    if (!c.exact) return start + wsum<4>(sq);
    double fv = start;     // (squares beyond n are exactly +0.0: no end-of-range test)
    for (int g8 = 0; 8 * g8 < c.n; ++g8)
        wpivots8<4, false>(sq, g8, [&](auto cs, auto q, double v) __attribute__((always_inline)) { fv += v; });
    return fv;
}
// scan (all waves), then the pick among the waves' candidates and, when asked, |u|^2 in index order
template <int C>
__device__ __forceinline__ int wscan(WgWave<C> &w, int &upper, bool with_fval)
{
    const WgCtx &c = w.c;
    const int lane = wg_lane(), kk = lane < c.W ? lane : 0;
    if (c.M32 != nullptr && !c.exact) {
        // fp32 screening pass first (see wg_scan32): its verdict stands only when it is certain
        if (w.profiling && lane == 0) reinterpret_cast<long long *>(wg_sm() + WgL<C>::prof)[16] += 1;
        wg_run(w, WG_SCAN32);
        WPROF_T0(w);
        if (with_fval) w.fval = und(wordered_norm2(w, w.soft));
        double sq[4];
        static_for<4>([&](auto cc) __attribute__((always_inline)) {
            const int j = lane + 64 * cc;
            const double uj = (j < c.n) ? SD(c, u)[j] : 0.0;
            sq[cc] = uj * uj;
        });
        // |fl32(M_r . u) - M_r . u| <= (K + 5) 2^-24 sum_j |M_rj u_j| for fma chains of at most K terms, two rounded operands per
        // term and a short tree of final additions; sum_j |M_rj u_j| <= |M_r| |u| and the rows are unit vectors.  K <= nquad.
        const double E = 1.001 * (double)(c.nquad + 8) * 5.9604644775390625e-08 * sqrt(wsum<4>(sq)) + 1e-300;
        const double *o = SD(c, cand) + 8 * kk;
        double bv = o[0];
        int bi = (lane < c.W) ? reinterpret_cast<const int *>(o + 5)[0] : kBig;
        int aux = lane;
        const int wbad = (lane < c.W) ? reinterpret_cast<const int *>(o + 6)[0] : 0;
        const double mq = (lane < c.W) ? o[2] : (double)DAQP_INF;
        wave_argmin(bv, bi, aux);
        const double other = (lane < c.W) ? ((lane == aux && bi != kBig) ? o[1] : o[0]) : (double)DAQP_INF;
        const double s2 = wave_min(other), minq = wave_min(mq);
        const bool anybad = __ballot(wbad != 0) != 0;
        const int wl = (bi == kBig) ? 0 : aux;
        const double q1 = rl(o[3], wl), gap1 = rl(o[4], wl);
        const int up1 = __builtin_amdgcn_readlane(reinterpret_cast<const int *>(o + 5)[1], wl);
        if (!anybad && minq >= E) { upper = 0; WPROF_ACC(w, 12); return kBig; }                  // certainly nothing violated
        if (!anybad && bi != kBig && bv + 2.0 * E < s2 && q1 < -E && gap1 > 2.0 * E) { upper = up1; WPROF_ACC(w, 12); return bi; }
        with_fval = false;      // (already done above)
        WPROF_ACC(w, 12);
        if (w.profiling && lane == 0) reinterpret_cast<long long *>(wg_sm() + WgL<C>::prof)[17] += 1;
    }
    wg_run(w, WG_SCAN);
    WPROF_T0(w);
    if (with_fval) w.fval = und(wordered_norm2(w, w.soft));
    // lane k <-> wave k's candidate; lowest value, then lowest row
    double bv = SD(c, cand)[8 * kk];
    int bi = (lane < c.W) ? reinterpret_cast<const int *>(SD(c, cand) + 8 * kk + 1)[0] : kBig;
    int bup = reinterpret_cast<const int *>(SD(c, cand) + 8 * kk + 1)[1];
    wave_argmin(bv, bi, bup);
    upper = bup;
    WPROF_ACC(w, 12);
    return bi;
}

// one step of iterative refinement on the active rows (auxiliary.c:498-593); rare, so the rows are read from the
// row-major scratch one lane per row
template <int C>
__device__ __forceinline__ void wrefine_active(WgWave<C> &w)
{
    const WgCtx &c = w.c;
    const int lane = wg_lane(), na = w.na, n = c.n;
    w.reuse = 0;
    double acc[C];
#pragma unroll
    for (int cc = 0; cc < C; ++cc) {
        const int i = lane + 64 * cc;
        acc[cc] = 0;
        if (i < na) {
            const int id = SI(c, ws)[i];
            const double *row = c.rowc + (size_t)SI(c, slot)[i] * c.ldr;
            double mu = 0;
            for (int j = (id < c.ms ? id : 0); j < n; ++j) mu += row[j] * SD(c, u)[j];
            const double d = (SI(c, sense)[id] & DAQP_LOWER) ? c.dlower[id] : c.dupper[id];
            acc[cc] = mu - d;
            if (SI(c, sense)[id] & DAQP_SOFT) acc[cc] -= w.stp->rho_soft * WLAMS(w)[i];
        }
    }
    wforward<C>(w, acc, 0);
    double b[C];
#pragma unroll
    for (int cc = 0; cc < C; ++cc) {
        const int i = lane + 64 * cc;
        b[cc] = 0;
        if (i < na) { SD(c, xl)[i] = acc[cc]; b[cc] = acc[cc] / SD(c, D)[i]; SD(c, zl)[i] = b[cc]; }
    }
    wbackward<C>(w, b, na);
#pragma unroll
    for (int cc = 0; cc < C; ++cc) {
        const int i = lane + 64 * cc;
        if (i < na) { SD(c, xl)[i] = b[cc]; WLAMS(w)[i] += b[cc]; }
    }
    WSYNC();
    double uu[C == 1 ? 2 : C + 1];   // columns: lane + 64 cc covers n <= 64 (C + 1) (n < cap <= 64 C)
    constexpr int CU = (C == 1 ? 2 : C + 1);
#pragma unroll
    for (int cc = 0; cc < CU; ++cc) { const int j = lane + 64 * cc; uu[cc] = (j < n) ? SD(c, u)[j] : 0.0; }
    for (int i = 0; i < na; ++i) {
        const double dl = SD(c, xl)[i];
        const int id = SI(c, ws)[i];
        const int j0 = id < c.ms ? id : 0;
        const double *row = c.rowc + (size_t)SI(c, slot)[i] * c.ldr;
#pragma unroll
        for (int cc = 0; cc < CU; ++cc) {
            const int j = lane + 64 * cc;
            if (j < n && j >= j0) uu[cc] -= row[j] * dl;
        }
    }
    WSYNC();
#pragma unroll
    for (int cc = 0; cc < CU; ++cc) { const int j = lane + 64 * cc; if (j < n) SD(c, u)[j] = uu[cc]; }
    WSYNC();
    w.fval = und(wordered_norm2(w, w.soft));
}

template <int C>
__device__ __forceinline__ void wreset_ws(WgWave<C> &w)
{
    const WgCtx &c = w.c;
    w.sing = kEmpty; w.na = 0; w.reuse = 0;
    w.use_w = 0;   // (an empty factor is an empty factor; what follows -- re-activation -- stays on the chains)
    // every slot of the scratch is free again, lowest on top
    for (int i = wg_lane(); i < c.cap; i += 64) SI(c, freestk)[i] = c.cap - 1 - i;
    w.nfree = c.cap; w.hi_slot = -1;
    WSYNC();
}

// ---------------------------------------------------------------------------------------------------------------------
// daqp_ldp (daqp.c:6-108) + daqp_activate_constraints (auxiliary.c:399-479) + daqp_pivot_last (auxiliary.c:379-396) as ONE
// explicit state machine, as in wave_ldp_reg.hip.h: the reference reaches add_constraint / remove_constraint from a dozen
// places and recurses through pivot_last; inlined, every primitive (and with it every parallel phase of the workgroup) was
// instantiated up to ten times -- 69 k lines of ISA, 780 bytes of scratch per lane.  Here each primitive and each command
// has exactly one call site and "who asked" is a continuation code.
//   mode 1: only rebuild the working set from the ACTIVE bits (tail of daqp_update_ldp).
// Returns the exit flag (mode 0) or the activation flag (mode 1).
// ---------------------------------------------------------------------------------------------------------------------
enum : int { WPC_START_LOOP, WPC_ITER, WPC_AFTER_DIR, WPC_SCAN, WPC_EDIT, WPC_ACT_BEGIN, WPC_ACT_NEXT, WPC_ACT_POST, WPC_ACT_POST2, WPC_DONE };
enum : int { WAFTER_NEXT_ITER, WAFTER_CYCLE_GUARD, WAFTER_ACT_POST };
enum : int { WACT_THEN_DONE, WACT_THEN_LOOP, WACT_THEN_NEXT_ITER, WACT_THEN_CYCLE_RESET };

template <int C>
__device__ __forceinline__ int wrun(WgWave<C> &w, int mode, bool need_activate, int &iterations)
{
    const WgCtx &c = w.c;
    const int lane = wg_lane();
    int flag = DAQP_EXIT_ITERLIMIT, it = 1, repaired = 0, stall = 0;
    double best = -1;
    const double fbound = 2 * w.stp->fval_bound;
    const bool timed = w.stp->time_limit > 0;
    int depth = 0, req_add = 1, req_id = 0, req_r = 0, after_edit = WAFTER_NEXT_ITER;
    double req_lam = 0;
    int act_then = WACT_THEN_DONE, act_blk = 0, act_i = 0, act_flag = 1;
    unsigned long long act_msk = 0;
    int tl_skip = 0, scan_first = 1, dir_then = WPC_AFTER_DIR, was_singular = 0;
    int pc;
    if (mode == 1 || need_activate) { wreset_ws(w); act_then = (mode == 1) ? WACT_THEN_DONE : WACT_THEN_LOOP; pc = WPC_ACT_BEGIN; }
    else pc = WPC_START_LOOP;
#define WTL_CHECK() (timed && !tl_skip && (it & 31) == 0 && time_is_up(w.t_start, w.stp->time_limit, w.tick_s))
#define WUNIFORM() do { w.na = uni(w.na); w.reuse = uni(w.reuse); w.sing = uni(w.sing); w.nfree = uni(w.nfree); w.hi_slot = uni(w.hi_slot); \
                        w.lam_b = uni(w.lam_b); w.overflow = uni(w.overflow); w.fast_na = uni(w.fast_na); } while (0)
    while (pc != WPC_DONE && !w.overflow) {
        // (belt and braces: the iterate's scalars are wave-uniform by construction; saying so once per state keeps every loop
        //  bounded by them a scalar loop whatever the optimizer concluded about the joins of the previous state)
        pc = uni(pc); it = uni(it); depth = uni(depth); req_add = uni(req_add); req_id = uni(req_id); req_r = uni(req_r); after_edit = uni(after_edit);
        WUNIFORM();
        WPROF_T0(w);
        switch (pc) {
        case WPC_START_LOOP:
            if (act_flag < 0) { flag = act_flag; pc = WPC_DONE; break; }
            it = 1;
            pc = (it < w.stp->iter_limit) ? WPC_ITER : WPC_DONE;
            break;
        // ---- lam* (CSP or singular direction): one site for the iteration and for the dependent-equality test of the activation
        case WPC_ITER:
            tl_skip = 0;
            was_singular = (w.sing != kEmpty);
            if (was_singular && dir_then == WPC_AFTER_DIR) wtrace(w, kTraceSingular);   // (the activation's dependency test is not an iteration)
            wdirection(w);
            WPROF_ACC(w, 0);
            pc = dir_then;
            break;
        case WPC_AFTER_DIR: {
            const int blk = wblocking_test(w);
            WPROF_ACC(w, 1);
            if (blk != kBig) { req_add = 0; req_r = blk; depth = 0; after_edit = WAFTER_NEXT_ITER; pc = WPC_EDIT; break; }
            if (was_singular) { flag = DAQP_EXIT_INFEASIBLE; pc = WPC_DONE; break; }
            wprimal(w);
            WPROF_ACC(w, 2);
            scan_first = 1;
            pc = WPC_SCAN;
            break;
        }
        // ---- feasibility scan: the iteration's (with |u|^2) or the one after refine_active
        case WPC_SCAN: {
            int upper = 0;
            const int pick = wscan(w, upper, scan_first != 0);
            WPROF_ACC(w, 3);
            if (scan_first) {
                if (ub(w.fval > fbound)) { flag = DAQP_EXIT_INFEASIBLE; pc = WPC_DONE; break; }
                after_edit = WAFTER_CYCLE_GUARD;
                if (pick == kBig) {
                    double dmin = SD(c, D)[0];
                    for (int i = 1; i < w.na; ++i) { const double di = SD(c, D)[i]; dmin = di < dmin ? di : dmin; }
                    dmin = und(dmin);
                    if (w.na > 2 && repaired != 1 && ub(dmin < w.stp->refactor_tol)) {
                        repaired = 1;
                        tl_skip = 1;
                        wtrace(w, kTraceRefactor);
                        const double *lam = WLAM(w);
                        for (int i = lane; i < w.na; i += 64) {
                            const int id = SI(c, ws)[i];
                            if (lam[i] >= 0) SI(c, sense)[id] &= ~DAQP_LOWER; else SI(c, sense)[id] |= DAQP_LOWER;
                        }
                        WSYNC();
                        wreset_ws(w);
                        act_then = WACT_THEN_NEXT_ITER; pc = WPC_ACT_BEGIN;
                        break;
                    }
                    if (w.na > 0 && ub(dmin < w.stp->pivot_tol)) {
                        wtrace(w, kTraceRefine);
                        wleave_w(w, w.na);
                        if (w.overflow) break;
                        wrefine_active(w);
                        scan_first = 0; after_edit = WAFTER_NEXT_ITER; tl_skip = 1;
                        pc = WPC_SCAN;
                        break;
                    }
                    flag = ub(w.soft > w.stp->primal_tol) ? DAQP_EXIT_SOFT_OPTIMAL : DAQP_EXIT_OPTIMAL;
                    pc = WPC_DONE;
                    break;
                }
            } else if (pick == kBig) {
                flag = ub(w.soft > w.stp->primal_tol) ? DAQP_EXIT_SOFT_OPTIMAL : DAQP_EXIT_OPTIMAL;
                pc = WPC_DONE;
                break;
            }
            // auxiliary.c:152-166: fix the side, lam <-> lam*, then add with multiplier +-1
            if (lane == 0) { if (upper) SI(c, sense)[pick] &= ~DAQP_LOWER; else SI(c, sense)[pick] |= DAQP_LOWER; }
            w.lam_b ^= 1;
            WSYNC();
            req_add = 1; req_id = pick; req_lam = upper ? 1.0 : -1.0; depth = 0; pc = WPC_EDIT;
            break;
        }
        // ---- add_constraint / remove_constraint with the pivot_last cascade (auxiliary.c:3-44, 379-396)
        case WPC_EDIT: {
            for (;;) {
                bool settled = false;
                WUNIFORM();
                req_add = uni(req_add); req_id = uni(req_id); req_r = uni(req_r); depth = uni(depth);
                if (req_add) { wpush_core(w, req_id, req_lam); if (w.overflow) break; WPROF_ACC(w, 4); }
                else { settled = wdrop_core(w, req_r) != 0; WPROF_ACC(w, 5); }     // a removal that left a singular factor does not pivot
                if (!settled) {
                    const int r = w.na - 2;
                    bool piv = false;
                    if (w.na > 1) {
                        const double dr = SD(c, D)[r], dlast = SD(c, D)[w.na - 1];
                        piv = ub(dr < w.stp->pivot_tol && dr < dlast);
                    }
                    if (piv) {
                        wleave_w(w, w.na);
                        if (w.overflow) break;
                        wtrace(w, kTracePivot);
                        if (lane == 0) { SI(c, pend_id)[depth] = SI(c, ws)[r]; SD(c, pend_lam)[depth] = WLAM(w)[r]; }
                        depth++;
                        WSYNC();
                        req_add = 0; req_r = r;
                        continue;
                    }
                    if (depth > 0 && w.sing == kEmpty) {
                        depth--;
                        req_id = uni(SI(c, pend_id)[depth]); req_lam = und(SD(c, pend_lam)[depth]);
                        req_add = 1;
                        continue;
                    }
                }
                break;
            }
            if (w.overflow) break;
            if (after_edit == WAFTER_ACT_POST) { pc = WPC_ACT_POST; break; }
            if (after_edit == WAFTER_CYCLE_GUARD) {   // daqp.c:66-85
                if (ub(w.fval - best < w.stp->progress_tol)) {
                    if (stall++ > w.stp->cycle_tol) {
                        if (repaired == 1) { flag = DAQP_EXIT_CYCLE; pc = WPC_DONE; break; }
                        repaired = 1;
                        wtrace(w, kTraceCycleReset);
                        wreset_ws(w);
                        act_then = WACT_THEN_CYCLE_RESET; pc = WPC_ACT_BEGIN;
                        break;
                    }
                } else { best = w.fval; stall = 0; }
            }
            if (WTL_CHECK()) { flag = DAQP_EXIT_TIMELIMIT; pc = WPC_DONE; break; }
            ++it;
            pc = (it < w.stp->iter_limit) ? WPC_ITER : WPC_DONE;   // falling out of the loop: flag stays ITERLIMIT
            break;
        }
        // ---- daqp_activate_constraints: ACTIVE rows in index order (auxiliary.c:399-479)
        case WPC_ACT_BEGIN:
            act_blk = 0; act_flag = 1;
            act_msk = __ballot(lane < c.m && (SI(c, sense)[lane < c.m ? lane : 0] & DAQP_ACTIVE));
            pc = WPC_ACT_NEXT;
            break;
        case WPC_ACT_NEXT: {
            while (act_msk == 0 && (act_blk + 1) * 64 < c.m) {
                act_blk++;
                const int r = act_blk * 64 + lane;
                act_msk = __ballot(r < c.m && (SI(c, sense)[r < c.m ? r : 0] & DAQP_ACTIVE));
            }
            if (act_msk == 0) {   // done: continue where the activation was requested from
                if (act_then == WACT_THEN_DONE) pc = WPC_DONE;
                else if (act_then == WACT_THEN_LOOP) pc = WPC_START_LOOP;
                else {
                    if (act_then == WACT_THEN_CYCLE_RESET) { stall = 0; best = -1; }
                    if (WTL_CHECK()) { flag = DAQP_EXIT_TIMELIMIT; pc = WPC_DONE; break; }
                    ++it;
                    pc = (it < w.stp->iter_limit) ? WPC_ITER : WPC_DONE;
                }
                break;
            }
            act_i = act_blk * 64 + __ffsll((long long)act_msk) - 1;
            act_msk &= act_msk - 1;
            req_add = 1; req_id = act_i; req_lam = ub((SI(c, sense)[act_i] & DAQP_LOWER) != 0) ? -1.0 : 1.0;
            depth = 0; after_edit = WAFTER_ACT_POST; pc = WPC_EDIT;
            break;
        }
        case WPC_ACT_POST: {
            if (w.sing == kEmpty) { pc = WPC_ACT_NEXT; break; }
            const int last = uni(SI(c, ws)[w.na - 1]);
            if (ub((SI(c, sense)[last] & DAQP_IMMUTABLE) != 0)) {   // a new equality depends on the active ones: its direction first
                dir_then = WPC_ACT_POST2;
                pc = WPC_ITER;
                break;
            }
            int fl = 1;
            for (int q = act_i; q < c.m; q += 1) {   // rows >= i: unactivated equalities are an error, the rest are cleaned
                const int sn = uni(SI(c, sense)[q]);
                if (sn & DAQP_ACTIVE) {
                    if (sn & DAQP_IMMUTABLE) fl = DAQP_EXIT_OVERDETERMINED_INITIAL;
                    else if (lane == 0) SI(c, sense)[q] = sn & ~DAQP_ACTIVE;
                }
            }
            if (lane == 0) SI(c, freestk)[w.nfree] = SI(c, slot)[w.na - 1];
            w.nfree++;
            w.na--;
            w.sing = kEmpty;
            act_flag = fl;
            WSYNC();
            act_msk = 0; act_blk = (c.m + 63) / 64;   // activation ends here (with act_flag)
            pc = WPC_ACT_NEXT;
            break;
        }
        case WPC_ACT_POST2: {   // consistent => ignore the dependent equality, else over-determined
            dir_then = WPC_AFTER_DIR;
            const int last = uni(SI(c, ws)[w.na - 1]);
            const double *lams = WLAMS(w);
            double resid = 0.0, scale = 1.0;
            for (int j = 0; j < w.na; ++j) {
                const int id = SI(c, ws)[j];
                const double bd = (SI(c, sense)[id] & DAQP_LOWER) ? c.dlower[id] : c.dupper[id];
                const double t = lams[j] * bd;
                resid += t;
                scale += t < 0 ? -t : t;
            }
            WSYNC();
            if (lane == 0) { SI(c, sense)[last] &= ~DAQP_ACTIVE; SI(c, freestk)[w.nfree] = SI(c, slot)[w.na - 1]; }
            w.nfree++;
            w.na--;
            w.sing = kEmpty;
            if (w.reuse > w.na) w.reuse = w.na;
            WSYNC();
            if (ub(resid <= w.stp->primal_tol * scale && resid >= -w.stp->primal_tol * scale)) { pc = WPC_ACT_NEXT; break; }
            act_flag = DAQP_EXIT_OVERDETERMINED_INITIAL;
            act_msk = 0; act_blk = (c.m + 63) / 64;
            pc = WPC_ACT_NEXT;
            break;
        }
        default:
            pc = WPC_DONE;
            break;
        }
    }
#undef WTL_CHECK
#undef WUNIFORM
    iterations = it;
    return (mode == 1) ? act_flag : flag;
}

} // namespace daqp_amd
