// tiny_ldp.hip.h -- the dual active-set iteration for TINY problems (n <= 12, m <= 48, working sets of at most 13 rows: MPC-size
// QPs, BASELINE config C3), SIXTEEN problems per wavefront: G = 4 lanes per problem.
//
// Why: with one wave per problem (wave_ldp_reg.hip.h) a 12 x 48 problem keeps <= 20 % of the lanes busy and the kernel is bound
// by instruction issue (round 2: 9.8 k wave instructions per problem).  Here one instruction stream serves 16 problems that walk
// the state machine of daqp_ldp (reference src/daqp.c:6-108) in LOCKSTEP PHASES -- activation step, CSP + ratio test, primal step +
// feasibility scan, row delete, row append, pivot / continuation logic -- each phase predicated per problem (EXEC mask), so a
// problem that removes a constraint and one that adds one share the pass.
//
// Where the state lives (per problem):
//   registers of its G lanes, lane s <-> constraint rows s, s+G, s+2G, ...: those rows of M themselves (12 x 12 doubles), their
//       d_upper, d_lower, -primal_tol*scaling and sense bits -- M is read from HBM once per solve, the feasibility scan
//       (auxiliary.c:89-198, the reference's dominant cost) is pure VALU with no cross-lane traffic until the final argmin;
//   registers, replicated in the G lanes: u, the iterate's scalars, the working set's ids / row-cache slots / flags bit-packed
//       in four 64-bit words (position-indexed, shifted by ALU on a removal);
//   LDS, element e of the problem at [e][problem]: the active-row cache (slot-indexed: a removal moves no row), packed L without
//       its unit diagonal, D, xldl, zldl, lam, lam*, and the CSP's right-hand side.  2 496 bytes per problem: 64 problems per CU.
//       The layout makes every address "per-lane base + immediate" and is bank-conflict free whatever element each problem touches.
// The small dense algebra (triangular solves over <= 13 positions) is replicated in the G lanes of a problem -- the lanes are there
// anyway -- with compile-time position indices; the parts that are sums over columns (Gram column, primal step) are split over the
// lanes and combined with DPP quad permutes.
//
// Arithmetic: FM = false keeps the reference's operation order everywhere (bit-identical results, -ffp-contract=off);
// FM = true (the library's default mode) fuses multiply-adds.
#pragma once
#include "wave_ldp_reg.hip.h"
#include "batch_dev.hip.h"

namespace daqp_amd {

constexpr int TNC = 12;     // columns
constexpr int TCAP = 13;    // working-set positions
constexpr int TMR = 48;     // constraint rows
// LDS elements of one problem
constexpr int kTR = 0, kTL = 156, kTV = 234, kTElems = 312;
constexpr int TV_D = 0, TV_XL = 13, TV_ZL = 26, TV_LA = 39, TV_LB = 52, TV_RHS = 65;
template <int G> struct TinyL {
    static constexpr int Q = 64 / G, RPL = TMR / G;
    static constexpr int bytes = kTElems * Q * 8;
};
__host__ __device__ constexpr int tlidx(int i, int j) { return i * (i - 1) / 2 + j; }   // L[i][j], i > j (no diagonal)
constexpr int kRowNone = 0x3fffffff;

// ---- cross-lane traffic inside a group of G lanes (G == 4: DPP quad permutes) --------------------------------------------
template <int G, int S> __device__ __forceinline__ int gbcast_i(int v)
{
    static_assert(G == 4, "groups of four lanes");
    return __builtin_amdgcn_update_dpp(v, v, S | (S << 2) | (S << 4) | (S << 6), 0xF, 0xF, false);
}
template <int G, int S> __device__ __forceinline__ double gbcast(double v)
{
    return __hiloint2double(gbcast_i<G, S>(__double2hiint(v)), gbcast_i<G, S>(__double2loint(v)));
}
template <int CTRL> __device__ __forceinline__ int dpp_i(int v) { return __builtin_amdgcn_update_dpp(v, v, CTRL, 0xF, 0xF, false); }
// sum of the group's four values as (s0 + s1) + (s2 + s3) -- the association of the reference's dot_row (factorization.c:4-15)
template <int G> __device__ __forceinline__ double gsum_pairs(double s)
{
    const double t = s + dpp_f64<0xB1>(s);      // lanes {0,1}: s0 + s1, lanes {2,3}: s2 + s3
    return t + dpp_f64<0x4E>(t);
}
template <int G> __device__ __forceinline__ int gor(int v) { v |= dpp_i<0xB1>(v); v |= dpp_i<0x4E>(v); return v; }
template <int G> __device__ __forceinline__ int gmin_i(int v)
{
    int o = dpp_i<0xB1>(v); v = o < v ? o : v;
    o = dpp_i<0x4E>(v); v = o < v ? o : v;
    return v;
}
// the value of the one lane of the group where `mine` holds (bitwise: exact), 0 if none
template <int G> __device__ __forceinline__ double gpick(double v, bool mine)
{
    const int lo = gor<G>(mine ? __double2loint(v) : 0), hi = gor<G>(mine ? __double2hiint(v) : 0);
    return __hiloint2double(hi, lo);
}

template <int G, int TRI, bool FM>
struct TW {
    static constexpr int Q = TinyL<G>::Q, RPL = TinyL<G>::RPL;
    double *sm;                 // this problem's LDS column: element e at sm[e * Q]
    int sub;                    // lane within the group
    // row view: rows sub + G*k
    double M[RPL][TNC];
    // d_upper, d_lower, scaling of the rows: read from HBM / L2 at the top of every scan (36 loads that complete behind the scan's
    // 132 multiply-adds) instead of 72 more registers held across the whole loop -- the register file is full: M alone is 264-288
    const double *gdu, *gdl, *gsc;   // (rebuilt from the problem index where they are used: tptr)
    double ep;                  // -primal_tol
    unsigned long long rs;      // 4 sense bits per own row (ACTIVE, LOWER, IMMUTABLE, SOFT)
    // replicated in the group
    double u[TNC];
    unsigned long long id0, id1, slw, flw;   // working set by position: ids (6 bits; 8 + 5), row-cache slots (4 bits), flags LOWER/IMMUTABLE/SOFT (3 bits)
    int n, m, ms, na, reuse, sing, lamsw, has_soft;
    unsigned slotmask;
    double fval, soft, dual_tol, sing_tol, pivot_tol, rho_soft;
    const DAQPSettings *stp;
    int *trace; int trace_cap, trace_len;
    // cycle probes (daqp_batch_enable_profile): per-wave sums by phase, wave-uniform
#ifdef DAQP_TINY_PROF
    bool prof; long long pt[12]; long long pt0;
#endif
};
#define TWT template <int G, int TRI, bool FM>
#define TWR TW<G, TRI, FM> &

// ---- the working set's packed words -----------------------------------------------------------------------------------
TWT __device__ __forceinline__ int tws_id(const TW<G, TRI, FM> &w, int i) { return i < 8 ? (int)((w.id0 >> (6 * i)) & 63) : (int)((w.id1 >> (6 * (i - 8))) & 63); }
TWT __device__ __forceinline__ int tws_slot(const TW<G, TRI, FM> &w, int i) { return (int)((w.slw >> (4 * i)) & 15); }
TWT __device__ __forceinline__ int tws_flag(const TW<G, TRI, FM> &w, int i) { return (int)((w.flw >> (3 * i)) & 7) << 1; }
TWT __device__ __forceinline__ void tws_set(TWR w, int i, int id, int slot, int fl)
{
    if (i < 8) w.id0 = (w.id0 & ~(63ull << (6 * i))) | ((unsigned long long)id << (6 * i));
    else w.id1 = (w.id1 & ~(63ull << (6 * (i - 8)))) | ((unsigned long long)id << (6 * (i - 8)));
    w.slw = (w.slw & ~(15ull << (4 * i))) | ((unsigned long long)slot << (4 * i));
    w.flw = (w.flw & ~(7ull << (3 * i))) | ((unsigned long long)((fl >> 1) & 7) << (3 * i));
}
__device__ __forceinline__ unsigned long long tpack_drop(unsigned long long v, int bits, int r)
{
    const unsigned long long lo = (1ull << (bits * r)) - 1ull;
    return (v & lo) | ((v >> bits) & ~lo);
}
TWT __device__ __forceinline__ void tws_remove(TWR w, int r)   // positions > r move down by one
{
    w.slw = tpack_drop(w.slw, 4, r);
    w.flw = tpack_drop(w.flw, 3, r);
    if (r < 8) {
        w.id0 = tpack_drop(w.id0, 6, r);
        w.id0 = (w.id0 & ~(63ull << 42)) | ((w.id1 & 63ull) << 42);
        w.id1 >>= 6;
    } else w.id1 = tpack_drop(w.id1, 6, r - 8);
}
TWT __device__ __forceinline__ void ttrace(TWR w, int ev)
{
    if (w.trace) {
        if (w.sub == 0 && w.trace_len < w.trace_cap) w.trace[w.trace_len] = ev;
        w.trace_len++;
    }
}
// sense bits of constraint id (group-uniform): only the owning lane holds them
TWT __device__ __forceinline__ void tsense_set(TWR w, int id, int set_bits, int clear_bits)
{
    if (w.sub == (id & (G - 1))) {
        const int sh = 4 * (id / G);
        w.rs = (w.rs | ((unsigned long long)set_bits << sh)) & ~((unsigned long long)clear_bits << sh);
    }
}
#ifdef DAQP_TINY_PROF
#define TPROF(w, slot) do { if ((w).prof) { const long long t1_ = (long long)__builtin_readcyclecounter(); (w).pt[slot] += t1_ - (w).pt0; (w).pt0 = t1_; } } while (0)
#else
#define TPROF(w, slot) do { } while (0)
#endif
#define TLAM(w) (kTV + ((w).lamsw ? TV_LB : TV_LA))
#define TLAMS(w) (kTV + ((w).lamsw ? TV_LA : TV_LB))

// wave-wide maximum as a wave-uniform value (call only where every lane is active)
__device__ __forceinline__ int wave_max_i(int v)
{
    int o = dpp_i<0xB1>(v); v = o > v ? o : v;
    o = dpp_i<0x4E>(v); v = o > v ? o : v;
    o = dpp_i<0x141>(v); v = o > v ? o : v;
    o = dpp_i<0x140>(v); v = o > v ? o : v;
    const int a = __builtin_amdgcn_readlane(v, 0), b = __builtin_amdgcn_readlane(v, 16), c = __builtin_amdgcn_readlane(v, 32), d = __builtin_amdgcn_readlane(v, 48);
    const int ab = a > b ? a : b, cd = c > d ? c : d;
    return ab > cd ? ab : cd;
}
// a value the optimizer cannot see through: compares derived from it are recomputed where they are used instead of being
// kept as SGPR-pair masks across the whole pass (the first version of this kernel spilled 500 SGPRs that way)
__device__ __forceinline__ int opaque(int v) { asm volatile("" : "+v"(v)); return v; }

// ---- forward substitution: x_i = rhs_i - sum_{j<i} L[i][j] x_j for rows i = from .. na-1, one row per trip, j ascending
// (auxiliary.c:316-339).  rhs_off: LDS vector the right-hand sides are read from (TV_RHS for the CSP; TV_XL itself for
// refine_active, which overwrites its residuals in place).  x goes to TV_XL, z = x / D to TV_ZL.
TWT __device__ __forceinline__ void tforward(TWR w, int from, int rhs_off)
{
    constexpr int Q = TW<G, TRI, FM>::Q;
    const int na = opaque(w.na);
    for (int i = from; i < na; ++i) {     // (after an add: one trip)
        double sum = w.sm[(kTV + rhs_off + i) * Q];
        const double *Lr = w.sm + (kTL + tlidx(i, 0)) * Q;
        double Lv[TCAP - 1], xv[TCAP - 1];
        static_for<TCAP - 1>([&](auto j) __attribute__((always_inline)) { Lv[j] = Lr[j * Q]; xv[j] = w.sm[(kTV + TV_XL + j) * Q]; });
        static_for<TCAP - 1>([&](auto j) __attribute__((always_inline)) {
            const double t = msub<FM>(sum, Lv[j], xv[j]);
            sum = (j < i) ? t : sum;
        });
        const double z = sum / w.sm[(kTV + TV_D + i) * Q];
        if (w.sub == 0) { w.sm[(kTV + TV_XL + i) * Q] = sum; w.sm[(kTV + TV_ZL + i) * Q] = z; }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    }
}
// b <- L' \ b over the leading positions, column-oriented: row j of L updates b_i, i < j, for j = cnt-1 .. 1; every b_i receives
// its subtractions in descending j, product b_j * L[j][i] (auxiliary.c:344-352).  b_j must be 0 beyond the problem's own count
// (those steps then subtract 0 * finite); cmax >= every problem's count is wave-uniform: rows beyond it are skipped by a scalar branch.
TWT __device__ __forceinline__ void tbackward(TWR w, double (&b)[TCAP], int cmax)
{
    constexpr int Q = TW<G, TRI, FM>::Q;
    static_for<TCAP - 1>([&](auto jj) __attribute__((always_inline)) {
        constexpr int j = TCAP - 1 - jj;                    // 12 .. 1
        if (j < cmax) {
            double Lv[j];
            static_for<j>([&](auto i) __attribute__((always_inline)) { Lv[i] = w.sm[(kTL + tlidx(j, i)) * Q]; });
            static_for<j>([&](auto i) __attribute__((always_inline)) { b[i] = msub<FM>(b[i], b[j], Lv[i]); });
        }
    });
}
TWT __device__ __forceinline__ void tstore_lams(TWR w, const double (&b)[TCAP])
{
    constexpr int Q = TW<G, TRI, FM>::Q;
    double *ls = w.sm + TLAMS(w) * Q;
    if (w.sub == 0) static_for<TCAP>([&](auto i) __attribute__((always_inline)) { ls[i * Q] = b[i]; });
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
}
// ratio test of auxiliary.c:277-311 (SOFT_WEIGHTS off) on lam* = b[]: returns the position to drop (or -1) after stepping
// lam towards lam*
TWT __device__ __forceinline__ int tblocking(TWR w, const double (&b)[TCAP])
{
    constexpr int Q = TW<G, TRI, FM>::Q;
    const int na = opaque(w.na);
    const bool regular = (w.sing == kEmpty);
    double *lm = w.sm + TLAM(w) * Q;
    unsigned blk = 0;
    static_for<TCAP>([&](auto i) __attribute__((always_inline)) {
        const int fl = (int)((w.flw >> (3 * i)) & 7) << 1;
        bool bl = (i < na) && !(fl & DAQP_IMMUTABLE);
        if (fl & DAQP_LOWER) { if (b[i] < w.dual_tol) bl = false; }
        else if (b[i] > -w.dual_tol) bl = false;
        blk |= bl ? (1u << i) : 0u;
    });
    if (blk == 0) return -1;
    double alpha = DAQP_INF;
    int rm = -1;
    for (unsigned rest = blk; rest; rest &= rest - 1) {     // candidates in ascending position, strict `<`: the first minimum wins
        const int i = __ffs((int)rest) - 1;
        double bi = 0;
        static_for<TCAP>([&](auto k) __attribute__((always_inline)) { bi = (k == i) ? b[k] : bi; });
        const double li = lm[i * Q];
        const double cand = regular ? -li / (bi - li) : -li / bi;
        if (cand < alpha) { alpha = cand; rm = i; }
    }
    if (rm < 0) return -1;
    if (w.sub == 0)
        static_for<TCAP>([&](auto i) __attribute__((always_inline)) {
            const double li = lm[i * Q];
            const double nv = regular ? li + alpha * (b[i] - li) : li + alpha * b[i];
            if (i < na) lm[i * Q] = nv;
        });
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    w.sing = kEmpty;
    return rm;
}
// u = -M_k' lam* in working-set order (auxiliary.c:46-88): lane s forms the components s, s+G, ... from the cached rows, then
// every lane of the group receives all of them.  lam*_i = 0 beyond the problem's working set (0 * a cached row: exact no-op);
// namax is wave-uniform.
TWT __device__ __forceinline__ void tprimal(TWR w, const double (&b)[TCAP], int namax)
{
    constexpr int Q = TW<G, TRI, FM>::Q, CP = TNC / G;
    double uu[CP];
    static_for<CP>([&](auto t) __attribute__((always_inline)) { uu[t] = 0.0; });
    static_for<TCAP>([&](auto i) __attribute__((always_inline)) {
        if (i < namax) {
            const double *row = w.sm + (kTR + (int)((w.slw >> (4 * i)) & 15) * TNC + w.sub) * Q;
            double rv[CP];
            static_for<CP>([&](auto t) __attribute__((always_inline)) { rv[t] = row[t * G * Q]; });
            static_for<CP>([&](auto t) __attribute__((always_inline)) { uu[t] = msub<FM>(uu[t], rv[t], b[i]); });
        }
    });
    static_for<TNC>([&](auto c) __attribute__((always_inline)) { w.u[c] = gbcast<G, c % G>(uu[c / G]); });
    double fv = 0;
    if (w.has_soft) {
        static_for<TCAP>([&](auto i) __attribute__((always_inline)) {
            if (i < w.na && (tws_flag(w, i) & DAQP_SOFT)) fv += b[i] * b[i];
        });
    }
    w.soft = fv * w.rho_soft;
}
// feasibility scan + most violated row (auxiliary.c:89-198): the rows of M from registers, k-ordered sums; returns the row
// (kRowNone: primal feasible) with its side, the bound on that side and its sense bits
TWT __device__ __forceinline__ int tscan(TWR w, int &upper, double &bound, int &sflags, bool with_fval)
{
    constexpr int RPL = TW<G, TRI, FM>::RPL, KG = 4;      // row slots per group: four independent k-ordered chains in flight
    static_assert(RPL % KG == 0, "row slots come in groups of four");
    if (with_fval) {
        double fv = w.soft;
        static_for<TNC>([&](auto j) __attribute__((always_inline)) { fv = madd<FM>(fv, w.u[j], w.u[j]); });
        w.fval = fv;
    }
    double bv = 0.0, bb = 0.0;
    int bi = kRowNone;      // row | side << 8 | sense << 12
    const int m = opaque(w.m);
    static_for<RPL / KG>([&](auto gg) __attribute__((always_inline)) {
        double mu[KG], du[KG], dl[KG], sc[KG];
        static_for<KG>([&](auto kk) __attribute__((always_inline)) {
            constexpr int k = KG * gg + kk;
            const int r = w.sub + G * k, rr = r < m ? r : 0;
            du[kk] = w.gdu[rr]; dl[kk] = w.gdl[rr]; sc[kk] = w.gsc[rr];
            mu[kk] = 0.0;
        });
        static_for<TNC>([&](auto j) __attribute__((always_inline)) {
            static_for<KG>([&](auto kk) __attribute__((always_inline)) {
                constexpr int k = KG * gg + kk;
                if constexpr (!(k < TRI && j < G * k)) mu[kk] = madd<FM>(mu[kk], w.M[k][j], w.u[j]);
            });
        });
        static_for<KG>([&](auto kk) __attribute__((always_inline)) {
            constexpr int k = KG * gg + kk;
            const int r = w.sub + G * k;
            const int sn = (int)((w.rs >> (4 * k)) & 15);
            const bool open = r < m && !(sn & (DAQP_ACTIVE + DAQP_IMMUTABLE));
            const double cu = du[kk] - mu[kk], cl = mu[kk] - dl[kk], bt = w.ep * sc[kk];
            const bool up = open && cu < bv && cu < bt;
            const bool lo = open && !up && cl < bv && cl < bt;
            bv = up ? cu : (lo ? cl : bv);
            bb = up ? du[kk] : (lo ? dl[kk] : bb);
            bi = up ? (r | 0x100 | (sn << 12)) : (lo ? (r | (sn << 12)) : bi);
        });
    });
    // group argmin: smaller value wins, the lower row on ties (the reference's strict `<` in index order)
    static_for<2>([&](auto s) __attribute__((always_inline)) {
        constexpr int ctrl = s == 0 ? 0xB1 : 0x4E;
        const double ov = dpp_f64<ctrl>(bv), ob = dpp_f64<ctrl>(bb);
        const int oi = dpp_i<ctrl>(bi);
        const bool take = (oi != kRowNone) && (bi == kRowNone || ov < bv || (ov == bv && (oi & 0xff) < (bi & 0xff)));
        bv = take ? ov : bv; bb = take ? ob : bb; bi = take ? oi : bi;
    });
    if (bi == kRowNone) return kRowNone;
    upper = (bi >> 8) & 1; bound = bb; sflags = (bi >> 12) & 15;
    return bi & 0xff;
}

// ---- LDL' row append (factorization.c:21-111): returns the new pivot D[na].  The row itself goes from its owner's registers
// into cache slot `slot`; sn = sense bits of the row; namax >= na of every appending problem is wave-uniform.
TWT __device__ __forceinline__ double tappend(TWR w, int id, int slot, int sn, int namax)
{
    constexpr int Q = TW<G, TRI, FM>::Q, RPL = TW<G, TRI, FM>::RPL;
    const int na = opaque(w.na), n = w.n;
    {   // the owning lane stores its registers
        double *dst = w.sm + (kTR + slot * TNC) * Q;
        const int kq = (w.sub == (id & (G - 1))) ? id / G : -1;
        static_for<RPL>([&](auto k) __attribute__((always_inline)) {
            if (opaque(kq) == k)     // (opaque: twelve separate guarded blocks -- merged into "store M[kq][j]" they send M to scratch memory)
                static_for<TNC>([&](auto j) __attribute__((always_inline)) {
                    if constexpr (k < TRI && j < G * k) dst[j * Q] = 0.0; else dst[j * Q] = w.M[k][j];
                });
        });
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    }
    TPROF(w, 7);
    w.sing = kEmpty;
    const int c0 = id < w.ms ? id : 0;
    // Gram column: g_k = M_{WS[k]} . M_id with the reference's four partial sums -- one per lane -- over the columns from
    // j = max(start columns), the n%4 tail on chain 0, then (s0 + s1) + (s2 + s3)
    auto dotrow = [&](int sk, int j) __attribute__((always_inline)) -> double {
        const int len = n - j, nfull = len >> 2;
        const double *ra = w.sm + (kTR + sk * TNC + j + w.sub) * Q, *rb = w.sm + (kTR + slot * TNC + j + w.sub) * Q;
        double av[3], bw[3];
        static_for<3>([&](auto t) __attribute__((always_inline)) { av[t] = ra[4 * t * Q]; bw[t] = rb[4 * t * Q]; });
        double s = 0.0;
        static_for<3>([&](auto t) __attribute__((always_inline)) {
            const double p = madd<FM>(s, av[t], bw[t]);
            s = (t < nfull) ? p : s;
        });
        if (len & 3) {      // (simple bounds only: start columns that are not multiples of 4)
            const double *ta = w.sm + (kTR + sk * TNC + j + 4 * nfull) * Q, *tb = w.sm + (kTR + slot * TNC + j + 4 * nfull) * Q;
            for (int e = 0; e < (len & 3); ++e) {
                const double p = madd<FM>(s, ta[e * Q], tb[e * Q]);
                s = (w.sub == 0) ? p : s;
            }
        }
        return gsum_pairs<G>(s);
    };
    double g[TCAP];
    static_for<TCAP - 1>([&](auto k) __attribute__((always_inline)) {
        g[k] = 0.0;
        if (k < namax) {      // (positions beyond the problem's own working set: slot 0 of the cache, a finite row; the value is never used)
            const int idk = (int)(k < 8 ? (w.id0 >> (6 * k)) & 63 : (w.id1 >> (6 * (k - 8))) & 63);
            const int j = (idk < w.ms) ? (c0 > idk ? c0 : idk) : c0;
            g[k] = dotrow((int)((w.slw >> (4 * k)) & 15), j);
        }
    });
    g[TCAP - 1] = 0.0;
    double dnew = dotrow(slot, c0);
    int ns_act = 0;
    if (w.has_soft) {
        static_for<TCAP>([&](auto k) __attribute__((always_inline)) { if (k < na && (tws_flag(w, k) & DAQP_SOFT)) ns_act++; });
        if (sn & DAQP_SOFT) { ns_act++; dnew += w.rho_soft; }
    }
    TPROF(w, 10);
    if (na == 0) return dnew;
    // l <- L \ g, column by column (every g_i receives its subtractions in ascending j: factorization.c:81-88); rows beyond
    // na carry unused values
    static_for<TCAP - 2>([&](auto j) __attribute__((always_inline)) {
        if (j < namax - 1) {
            constexpr int jc = j;
            double Lv[TCAP - 1 - jc];
            static_for<TCAP - 1 - jc>([&](auto ii) __attribute__((always_inline)) { Lv[ii] = w.sm[(kTL + tlidx(jc + 1 + ii, jc)) * Q]; });
            static_for<TCAP - 1 - jc>([&](auto ii) __attribute__((always_inline)) { g[jc + 1 + ii] = msub<FM>(g[jc + 1 + ii], Lv[ii], g[jc]); });
        }
    });
    // l_k /= D_k, d_new -= sum_k l_k^2 D_k in k order (factorization.c:93-103): lane s divides for k = s, s+G, ...
    constexpr int KP = (TCAP + G - 1) / G;
    double p[KP];
    static_for<KP>([&](auto c) __attribute__((always_inline)) {
        const int k = w.sub + G * c;
        double t = 0.0;
        static_for<G>([&](auto s) __attribute__((always_inline)) { if constexpr (G * c + s < TCAP) t = (w.sub == s) ? g[G * c + s] : t; });
        p[c] = 0.0;
        if (k < na) {
            const double lk = t / w.sm[(kTV + TV_D + k) * Q];
            w.sm[(kTL + tlidx(na, 0) + k) * Q] = lk;
            p[c] = t * lk;
        }
    });
    double acc = dnew;
    if constexpr (FM) {
        double s = 0.0;
        static_for<KP>([&](auto c) __attribute__((always_inline)) { s += p[c]; });
        acc -= gsum_pairs<G>(s);
    } else {
        static_for<TCAP>([&](auto k) __attribute__((always_inline)) { acc -= gbcast<G, k % G>(p[k / G]); });   // (p_k = 0 beyond na)
    }
    if (acc < w.sing_tol || na >= n + ns_act) { w.sing = na; acc = 0.0; }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    return acc;
}

// auxiliary.c:27-40: append constraint id (sense bits sn, multiplier lamv, CSP right-hand side rhs = -bound)
TWT __device__ __forceinline__ void tpush(TWR w, int id, int sn, double lamv, double rhs, int namax)
{
    constexpr int Q = TW<G, TRI, FM>::Q;
    ttrace(w, id + 1);
    tsense_set(w, id, DAQP_ACTIVE, 0);
    const int slot = __ffs((int)~w.slotmask) - 1;
    w.slotmask |= 1u << slot;
    const double dnew = tappend(w, id, slot, sn | DAQP_ACTIVE, namax);
    tws_set(w, w.na, id, slot, sn);
    if (w.sub == 0) {
        w.sm[(TLAM(w) + w.na) * Q] = lamv;
        w.sm[(kTV + TV_D + w.na) * Q] = dnew;
        w.sm[(kTV + TV_RHS + w.na) * Q] = rhs;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    w.na++;
}

// ---- LDL' row delete (factorization.c:112-151) + the bookkeeping of auxiliary.c:3-22; returns 1 if the factor became singular.
// numax >= na - r - 1 of every dropping problem is wave-uniform.
TWT __device__ __forceinline__ int tdrop(TWR w, int r, int numax)
{
    constexpr int Q = TW<G, TRI, FM>::Q;
    const int na = opaque(w.na);
    const int idr = tws_id(w, r);
    ttrace(w, -(idr + 1));
    tsense_set(w, idr, 0, DAQP_ACTIVE);
    w.slotmask &= ~(1u << tws_slot(w, r));
    const int nupd = na - r - 1;
    double *lm = w.sm + TLAM(w) * Q;
    if (numax > 0) {
        constexpr int WP = (TCAP - 1 + G - 1) / G;
        // column r below the diagonal: lane s keeps w_t for t = s, s+G, ... (0 beyond the problem's own rows)
        double wv[WP];
        static_for<WP>([&](auto c) __attribute__((always_inline)) {
            const int t = w.sub + G * c;
            const double l = w.sm[(kTL + tlidx(t < nupd ? r + 1 + t : 1, t < nupd ? r : 0)) * Q];
            wv[c] = (t < nupd) ? l : 0.0;
        });
        // rows r+1.. move up by one and lose column r: destination row i' reads source row i'+1 (a higher address), rows in
        // ascending order, each row's loads before its stores; lane s moves the columns s, s+G, ...
        static_for<TCAP - 2>([&](auto ii) __attribute__((always_inline)) {
            constexpr int i = ii + 1;                       // destination rows 1 .. 11
            if (i >= r && i < na - 1) {
                constexpr int CPR = (i + G - 1) / G;
                double tmp[CPR];
                static_for<CPR>([&](auto c) __attribute__((always_inline)) {
                    const int j = w.sub + G * c;
                    const int js = (j < i) ? j + (j >= r ? 1 : 0) : 0;
                    tmp[c] = w.sm[(kTL + tlidx(i + 1, 0) + js) * Q];
                });
                static_for<CPR>([&](auto c) __attribute__((always_inline)) {
                    const int j = w.sub + G * c;
                    if (j < i) w.sm[(kTL + tlidx(i, 0) + j) * Q] = tmp[c];
                });
            }
        });
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        // Gill-Golub-Murray-Saunders C1 rank-one update of the trailing block, pivot by pivot.  Problems whose own update is
        // shorter run the remaining steps with p = 0 and store nothing.
        double alpha = w.sm[(kTV + TV_D + r) * Q];
        static_for<TCAP - 1>([&](auto j) __attribute__((always_inline)) {
            if (j < numax) {
                const bool mine = j < nupd;
                const double p = gbcast<G, j % G>(wv[j / G]);
                const double Di = w.sm[(kTV + TV_D + (mine ? r + 1 + j : 0)) * Q];
                const double dbar = Di + alpha * p * p;
                // beta = p*alpha/dbar and alpha' = D_i*alpha/dbar: two divisions by the same number, one in the even and one
                // in the odd lanes
                const double num = (w.sub & 1) ? Di * alpha : p * alpha;
                const double quo = num / dbar;
                const double beta = gbcast<G, 0>(quo);
                alpha = gbcast<G, 1>(quo);
                if (mine && w.sub == 0) w.sm[(kTV + TV_D + r + j) * Q] = dbar;
                static_for<WP>([&](auto c) __attribute__((always_inline)) {
                    if constexpr (G * c + G - 1 > j) {
                        const int t = w.sub + G * c;
                        if (t > j && t < nupd) {
                            double *Lp = w.sm + (kTL + tlidx(r + t, r + j)) * Q;
                            const double l = *Lp;
                            wv[c] = msub<FM>(wv[c], p, l);
                            *Lp = madd<FM>(l, beta, wv[c]);
                        }
                    }
                });
            }
        });
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    }
    // the working-set vectors close the gap: lam and the CSP's right-hand sides (D was rewritten by the update; x, z are
    // recomputed from `reuse`)
    {
        constexpr int VP = (TCAP - 1 + G - 1) / G;
        double *rh = w.sm + (kTV + TV_RHS) * Q;
        double tl[VP], tr[VP];
        static_for<VP>([&](auto c) __attribute__((always_inline)) {
            const int i = w.sub + G * c;
            const int s = (i >= r && i < na - 1) ? i + 1 : 0;
            tl[c] = lm[s * Q]; tr[c] = rh[s * Q];
        });
        static_for<VP>([&](auto c) __attribute__((always_inline)) {
            const int i = w.sub + G * c;
            if (i >= r && i < na - 1) { lm[i * Q] = tl[c]; rh[i * Q] = tr[c]; }
        });
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    }
    tws_remove(w, r);
    w.na = na - 1;
    if (r < w.reuse) w.reuse = r;
    if (w.na > 0 && w.sm[(kTV + TV_D + w.na - 1) * Q] < w.sing_tol) {
        w.sing = w.na - 1;
        if (w.sub == 0) w.sm[(kTV + TV_D + w.na - 1) * Q] = 0.0;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        return 1;
    }
    return 0;
}

// sense bits of constraint id (from the owning lane) and its bound on the side the LOWER bit says
TWT __device__ __forceinline__ void trow_lookup(TWR w, int id, double &bound, int &sn)
{
    const bool owner = w.sub == (id & (G - 1));
    const int s = (int)((w.rs >> (4 * (id / G))) & 15);
    sn = gor<G>(owner ? s : 0);
    bound = (sn & DAQP_LOWER) ? w.gdl[id] : w.gdu[id];
}

// ---------------------------------------------------------------------------------------------------------------------------
// daqp_ldp (daqp.c:6-108) + daqp_activate_constraints (auxiliary.c:399-479) + daqp_pivot_last (auxiliary.c:379-396) for the
// problems of a wave, in lockstep passes.  Per problem (TCtl): st = what it does next; an edit request (add / drop) with its
// continuation; an activation cursor.  Every primitive has ONE call site (the triangular solves serve the CSP, the singular
// direction and refine_active alike: `kind` says which), so a pass is one straight walk through the code.  The wave is
// persistent: a problem that finishes is retired and its four lanes take the next problem off the batch's counter
// (tiny_kernel.hip.h), so the passes of a wave are not bounded by the slowest of sixteen problems.
// ---------------------------------------------------------------------------------------------------------------------------
enum : int { TST_ACT, TST_ITER, TST_EDIT, TST_DONE, TST_EMPTY };
enum : int { TK_CSP, TK_SING, TK_REFINE, TK_ACTSING };    // what the solve phase of a TST_ITER problem computes
enum : int { TAFTER_NEXT_ITER, TAFTER_CYCLE_GUARD, TAFTER_ACT_POST };
enum : int { TACT_THEN_DONE, TACT_THEN_LOOP, TACT_THEN_NEXT_ITER, TACT_THEN_CYCLE_RESET };

struct TCtl {
    int st, kind, it, flag, repaired, stall, tl_skip;
    int depth, req_add, req_id, req_r, req_sn, after;
    int act_then, act_i, act_next, act_flag;
    double best, req_lam, req_rhs;
};
// wave-uniform loop invariants (settings)
struct TSet {
    double fbound, progress_tol, time_limit, refactor_tol, primal_tol, tick_s;
    int iter_limit, cycle_tol, mode;
    unsigned long long *tstart;
    double *pend;       // [N][13][3]
};

#define TBEGIN_ITER(w, c, S) do { (c).st = ((c).it < (S).iter_limit) ? TST_ITER : TST_DONE; (c).kind = ((w).sing != kEmpty) ? TK_SING : TK_CSP; (c).tl_skip = 0; } while (0)

// a fresh problem in these lanes (mode 1: only rebuild the working set from the ACTIVE bits)
TWT __device__ __forceinline__ void tstart(TWR w, TCtl &c, const TSet &S, bool need_activate)
{
    c.flag = DAQP_EXIT_ITERLIMIT; c.it = 1; c.repaired = 0; c.stall = 0; c.best = -1; c.tl_skip = 0;
    c.depth = 0; c.req_add = 1; c.req_id = 0; c.req_r = 0; c.req_sn = 0; c.after = TAFTER_NEXT_ITER; c.req_lam = 0; c.req_rhs = 0;
    c.act_then = TACT_THEN_DONE; c.act_i = 0; c.act_next = 0; c.act_flag = 1; c.kind = TK_CSP;
    if (S.mode == 1 || need_activate) {
        w.sing = kEmpty; w.na = 0; w.reuse = 0; w.slotmask = 0;
        c.act_then = (S.mode == 1) ? TACT_THEN_DONE : TACT_THEN_LOOP;
        c.st = TST_ACT;
    } else TBEGIN_ITER(w, c, S);
}

// one lockstep pass: every live problem of the wave advances by one step of its state machine
TWT __device__ __forceinline__ void tpass(TWR w, TCtl &c, const TSet &S, int q)
{
    constexpr int Q = TW<G, TRI, FM>::Q, RPL = TW<G, TRI, FM>::RPL;
#define TTL_CHECK() (S.tstart != nullptr && !c.tl_skip && (c.it & 31) == 0 && time_is_up(S.tstart[q], S.time_limit, S.tick_s))
    const bool live = c.st != TST_DONE && c.st != TST_EMPTY;
    // wave-uniform bounds of this pass's position loops
    const int namax = wave_max_i(live ? w.na : 0);
#ifdef DAQP_TINY_PROF
    if (w.prof) { w.pt0 = (long long)__builtin_readcyclecounter(); w.pt[11] += 1; }
#endif
    // ---- activation cursor: the next ACTIVE-marked row in index order (auxiliary.c:399-479)
    if (c.st == TST_ACT) {
        int cand = kRowNone;
        static_for<RPL>([&](auto kk) __attribute__((always_inline)) {
            constexpr int k = RPL - 1 - kk;
            const int r = w.sub + G * k;
            if (r >= c.act_next && r < w.m && ((w.rs >> (4 * k)) & DAQP_ACTIVE)) cand = r;
        });
        cand = gmin_i<G>(cand);
        if (cand == kRowNone) {
            if (c.act_then == TACT_THEN_DONE) c.st = TST_DONE;
            else if (c.act_then == TACT_THEN_LOOP) {
                if (c.act_flag < 0) { c.flag = c.act_flag; c.st = TST_DONE; }
                else { c.it = 1; TBEGIN_ITER(w, c, S); }
            } else {
                if (c.act_then == TACT_THEN_CYCLE_RESET) { c.stall = 0; c.best = -1; }
                if (TTL_CHECK()) { c.flag = DAQP_EXIT_TIMELIMIT; c.st = TST_DONE; }
                else { ++c.it; TBEGIN_ITER(w, c, S); }
            }
        } else {
            double bd; int sn;
            trow_lookup(w, cand, bd, sn);
            c.act_i = cand;
            c.req_add = 1; c.req_id = cand; c.req_sn = sn; c.req_lam = (sn & DAQP_LOWER) ? -1.0 : 1.0; c.req_rhs = -bd;
            c.depth = 0; c.after = TAFTER_ACT_POST; c.st = TST_EDIT;
        }
    }
    TPROF(w, 0);
    // ---- the solve phase of an iteration (daqp.c:12-21, 86-93): CSP (auxiliary.c:314-354) or singular direction
    // (auxiliary.c:357-376), or the solves of refine_active (auxiliary.c:498-593), through the same two substitutions
    double b[TCAP];
    bool go_primal = false, go_scan = false;
    if (c.st == TST_ITER) {
        const int kind = c.kind;
        const bool dir = kind == TK_SING || kind == TK_ACTSING;
        if (kind == TK_SING) ttrace(w, kTraceSingular);
        if (kind == TK_REFINE) {    // residuals of the active rows into xldl (auxiliary.c:505-541)
            w.reuse = 0;
            const double *ls = w.sm + TLAMS(w) * Q;
            for (int i = 0; i < w.na; ++i) {
                const int id = tws_id(w, i), fl = tws_flag(w, i);
                const double *row = w.sm + (kTR + tws_slot(w, i) * TNC) * Q;
                const int j0 = id < w.ms ? id : 0;
                double mu = 0;
                static_for<TNC>([&](auto j) __attribute__((always_inline)) {
                    const double t = mu + row[j * Q] * w.u[j];
                    mu = (j >= j0 && j < w.n) ? t : mu;
                });
                double res = mu - (-w.sm[(kTV + TV_RHS + i) * Q]);      // d = -rhs exactly
                if (fl & DAQP_SOFT) res -= w.rho_soft * ls[i * Q];
                if (w.sub == 0) w.sm[(kTV + TV_XL + i) * Q] = res;
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        }
        if (!dir) tforward(w, kind == TK_REFINE ? 0 : w.reuse, kind == TK_REFINE ? TV_XL : TV_RHS);
        TPROF(w, 1);
        if (dir) {
            const int s = w.sing;
            const double *Ls = w.sm + (kTL + tlidx(s, 0)) * Q;
            static_for<TCAP>([&](auto i) __attribute__((always_inline)) {
                const double l = Ls[(i < TCAP - 1 ? i : 0) * Q];
                b[i] = (i < s) ? -l : 0.0;
            });
        } else {
            const int na = opaque(w.na);
            static_for<TCAP>([&](auto i) __attribute__((always_inline)) {
                const double z = w.sm[(kTV + TV_ZL + i) * Q];
                b[i] = (i < na) ? z : 0.0;
            });
        }
        tbackward(w, b, namax);
        TPROF(w, 2);
        if (dir) {
            const int s = w.sing;
            const bool flip = (tws_flag(w, s) & DAQP_LOWER) != 0;
            static_for<TCAP>([&](auto i) __attribute__((always_inline)) {
                const double v = (i == s) ? 1.0 : b[i];
                b[i] = (i <= s) ? (flip ? -v : v) : 0.0;
            });
        }
        if (kind == TK_REFINE) {    // auxiliary.c:568-592: lam* += delta, u -= M' delta, fval
            const int na = w.na;
            const double *ls = w.sm + TLAMS(w) * Q;
            for (int i = 0; i < na; ++i) {
                const int id = tws_id(w, i);
                const double *row = w.sm + (kTR + tws_slot(w, i) * TNC) * Q;
                const int j0 = id < w.ms ? id : 0;
                double di = 0;
                static_for<TCAP>([&](auto k) __attribute__((always_inline)) { di = (k == i) ? b[k] : di; });
                static_for<TNC>([&](auto j) __attribute__((always_inline)) {
                    const double t = w.u[j] - row[j * Q] * di;
                    w.u[j] = (j >= j0 && j < w.n) ? t : w.u[j];
                });
            }
            static_for<TCAP>([&](auto i) __attribute__((always_inline)) {
                if (i < na) {
                    if (w.sub == 0) w.sm[(kTV + TV_XL + i) * Q] = b[i];
                    b[i] = ls[i * Q] + b[i];
                }
            });
            double fv = w.soft;
            static_for<TNC>([&](auto j) __attribute__((always_inline)) { fv += w.u[j] * w.u[j]; });
            w.fval = fv;
            tstore_lams(w, b);
            go_scan = true;
        } else {
            if (kind == TK_CSP) w.reuse = w.na;
            tstore_lams(w, b);
            if (kind == TK_ACTSING) {   // auxiliary.c:424-459: a new equality depends on the active ones: consistent => ignore it
                const int last = tws_id(w, w.na - 1);
                double resid = 0.0, scale = 1.0;
                static_for<TCAP>([&](auto j) __attribute__((always_inline)) {
                    if (j < w.na) {
                        const double t = b[j] * (-w.sm[(kTV + TV_RHS + j) * Q]);
                        resid += t;
                        scale += t < 0 ? -t : t;
                    }
                });
                tsense_set(w, last, 0, DAQP_ACTIVE);
                w.slotmask &= ~(1u << tws_slot(w, w.na - 1));
                w.na--;
                w.sing = kEmpty;
                if (w.reuse > w.na) w.reuse = w.na;
                if (resid <= S.primal_tol * scale && resid >= -S.primal_tol * scale) c.act_next = c.act_i + 1;
                else { c.act_flag = DAQP_EXIT_OVERDETERMINED_INITIAL; c.act_next = kRowNone; }
                c.st = TST_ACT;
            } else {
                const int blk = tblocking(w, b);
                if (blk >= 0) { c.req_add = 0; c.req_r = blk; c.depth = 0; c.after = TAFTER_NEXT_ITER; c.st = TST_EDIT; }
                else if (kind == TK_SING) { c.flag = DAQP_EXIT_INFEASIBLE; c.st = TST_DONE; }
                else go_primal = true;
            }
        }
    }
    TPROF(w, 3);
    // ---- primal step, feasibility scan and what follows from it (daqp.c:22-64)
    if (go_primal || go_scan) {
        if (go_primal) tprimal(w, b, namax);
        TPROF(w, 4);
        int upper = 0, sn = 0;
        double bd = 0;
        const int pick = tscan(w, upper, bd, sn, go_primal);
        TPROF(w, 5);
        if (go_primal && w.fval > S.fbound) { c.flag = DAQP_EXIT_INFEASIBLE; c.st = TST_DONE; }
        else if (pick == kRowNone) {
            bool fin = true;
            if (go_primal) {
                double dmin = DAQP_INF;
                const int na = opaque(w.na);
                static_for<TCAP>([&](auto i) __attribute__((always_inline)) {
                    const double d = w.sm[(kTV + TV_D + i) * Q];
                    dmin = (i < na && d < dmin) ? d : dmin;
                });
                if (w.na > 2 && c.repaired != 1 && dmin < S.refactor_tol) {     // daqp.c:33-46
                    c.repaired = 1; c.tl_skip = 1;
                    ttrace(w, kTraceRefactor);
                    const double *lm = w.sm + TLAM(w) * Q;
                    for (int i = 0; i < w.na; ++i) {
                        const int id = tws_id(w, i);
                        if (lm[i * Q] >= 0) tsense_set(w, id, 0, DAQP_LOWER); else tsense_set(w, id, DAQP_LOWER, 0);
                    }
                    w.sing = kEmpty; w.na = 0; w.reuse = 0; w.slotmask = 0;
                    c.act_then = TACT_THEN_NEXT_ITER; c.act_next = 0; c.act_flag = 1; c.st = TST_ACT;
                    fin = false;
                } else if (w.na > 0 && dmin < w.pivot_tol) {                         // daqp.c:52-56: refine, then scan again
                    ttrace(w, kTraceRefine);
                    c.kind = TK_REFINE; c.tl_skip = 1;
                    fin = false;
                }
            }
            if (fin) {
                c.flag = (w.soft > S.primal_tol) ? DAQP_EXIT_SOFT_OPTIMAL : DAQP_EXIT_OPTIMAL;
                c.st = TST_DONE;
            }
        } else {
            // auxiliary.c:152-166: fix the side, lam <-> lam*, then add with multiplier +-1
            if (upper) tsense_set(w, pick, 0, DAQP_LOWER); else tsense_set(w, pick, DAQP_LOWER, 0);
            w.lamsw ^= 1;
            c.req_add = 1; c.req_id = pick; c.req_sn = upper ? (sn & ~DAQP_LOWER) : (sn | DAQP_LOWER);
            c.req_lam = upper ? 1.0 : -1.0; c.req_rhs = -bd;
            c.depth = 0; c.after = go_scan ? TAFTER_NEXT_ITER : TAFTER_CYCLE_GUARD; c.st = TST_EDIT;
        }
    }
    TPROF(w, 6);
    // ---- the edit: remove_constraint / add_constraint (auxiliary.c:3-44), one per problem and pass
    const bool dropping = c.st == TST_EDIT && !c.req_add, pushing = c.st == TST_EDIT && c.req_add;
    bool edited = false, sing_after_drop = false;
    if (__any(dropping)) {
        const int numax = wave_max_i(dropping ? w.na - c.req_r - 1 : 0);
        if (dropping) { sing_after_drop = tdrop(w, c.req_r, numax) != 0; edited = true; }
    }
    TPROF(w, 7);
    if (__any(pushing)) {
        const int pmax = wave_max_i(pushing ? w.na : 0);
        if (pushing) { tpush(w, c.req_id, c.req_sn, c.req_lam, c.req_rhs, pmax); edited = true; }
    }
    TPROF(w, 8);
    // ---- daqp_pivot_last (auxiliary.c:379-396) as a stack of pending re-insertions, then the requester's continuation
    if (edited) {
        bool more = false;
        if (!sing_after_drop) {
            const int r = w.na - 2;
            bool piv = false;
            if (w.na > 1) {
                const double dr = w.sm[(kTV + TV_D + r) * Q], dlast = w.sm[(kTV + TV_D + w.na - 1) * Q];
                piv = dr < w.pivot_tol && dr < dlast;
            }
            if (piv) {
                ttrace(w, kTracePivot);
                double *pend = S.pend + (size_t)q * 3 * TCAP;
                if (w.sub == 0) {
                    pend[3 * c.depth] = __hiloint2double(tws_flag(w, r), tws_id(w, r));
                    pend[3 * c.depth + 1] = w.sm[(TLAM(w) + r) * Q];
                    pend[3 * c.depth + 2] = w.sm[(kTV + TV_RHS + r) * Q];
                }
                c.depth++;
                c.req_add = 0; c.req_r = r; more = true;
            } else if (c.depth > 0 && w.sing == kEmpty) {
                c.depth--;
                const double *pend = S.pend + (size_t)q * 3 * TCAP;
                const double key = gpick<G>(w.sub == 0 ? pend[3 * c.depth] : 0.0, w.sub == 0);
                c.req_id = __double2loint(key); c.req_sn = __double2hiint(key);
                c.req_lam = gpick<G>(w.sub == 0 ? pend[3 * c.depth + 1] : 0.0, w.sub == 0);
                c.req_rhs = gpick<G>(w.sub == 0 ? pend[3 * c.depth + 2] : 0.0, w.sub == 0);
                c.req_add = 1; more = true;
            }
        }
        if (!more) {
            bool next_iter = false;
            if (c.after == TAFTER_ACT_POST) {
                if (w.sing == kEmpty) { c.act_next = c.act_i + 1; c.st = TST_ACT; }
                else if (tws_flag(w, w.na - 1) & DAQP_IMMUTABLE) { c.st = TST_ITER; c.kind = TK_ACTSING; }
                else {
                    int bad = 0;
                    static_for<RPL>([&](auto k) __attribute__((always_inline)) {   // rows >= act_i: unactivated equalities are an error, the rest are cleaned
                        const int rr = w.sub + G * k;
                        const int sn = (int)((w.rs >> (4 * k)) & 15);
                        const bool later = rr >= c.act_i && rr < w.m && (sn & DAQP_ACTIVE);
                        if (later && (sn & DAQP_IMMUTABLE)) bad = 1;
                        if (later && !(sn & DAQP_IMMUTABLE)) w.rs &= ~((unsigned long long)DAQP_ACTIVE << (4 * k));
                    });
                    bad = gor<G>(bad);
                    w.slotmask &= ~(1u << tws_slot(w, w.na - 1));
                    w.na--;
                    w.sing = kEmpty;
                    c.act_flag = bad ? DAQP_EXIT_OVERDETERMINED_INITIAL : 1;
                    c.act_next = kRowNone;
                    c.st = TST_ACT;
                }
            } else if (c.after == TAFTER_CYCLE_GUARD) {   // daqp.c:66-85
                next_iter = true;
                if (w.fval - c.best < S.progress_tol) {
                    if (c.stall++ > S.cycle_tol) {
                        if (c.repaired == 1) { c.flag = DAQP_EXIT_CYCLE; c.st = TST_DONE; next_iter = false; }
                        else {
                            c.repaired = 1;
                            ttrace(w, kTraceCycleReset);
                            w.sing = kEmpty; w.na = 0; w.reuse = 0; w.slotmask = 0;
                            c.act_then = TACT_THEN_CYCLE_RESET; c.act_next = 0; c.act_flag = 1; c.st = TST_ACT;
                            next_iter = false;
                        }
                    }
                } else { c.best = w.fval; c.stall = 0; }
            } else next_iter = true;
            if (next_iter) {
                if (TTL_CHECK()) { c.flag = DAQP_EXIT_TIMELIMIT; c.st = TST_DONE; }
                else { ++c.it; TBEGIN_ITER(w, c, S); }
            }
        }
    }
    TPROF(w, 9);
#undef TTL_CHECK
}

#undef TWT
#undef TWR
} // namespace daqp_amd
