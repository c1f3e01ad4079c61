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
    const double *gdu, *gdl, *gsc;
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
#define TLAM(w) (kTV + ((w).lamsw ? TV_LB : TV_LA))
#define TLAMS(w) (kTV + ((w).lamsw ? TV_LA : TV_LB))

// ---- forward substitution: x_i = rhs_i - sum_{j<i} L[i][j] x_j for rows i = from .. na-1, one row per trip, j ascending
// (auxiliary.c:316-339).  rhs_off: LDS vector the right-hand sides are read from (TV_RHS for the CSP; TV_XL itself for
// refine_active, which overwrites its residuals in place).  x goes to TV_XL, z = x / D to TV_ZL.
TWT __device__ __forceinline__ void tforward(TWR w, int from, int rhs_off)
{
    constexpr int Q = TW<G, TRI, FM>::Q;
    for (int i = from; i < w.na; ++i) {     // (after an add: one trip)
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
// b <- L' \ b over the leading cnt positions, column-oriented: row j of L updates b_i, i < j, for j = cnt-1 .. 1; every b_i
// receives its subtractions in descending j, product b_j * L[j][i] (auxiliary.c:344-352).  b_i must be 0 for i >= cnt.
TWT __device__ __forceinline__ void tbackward(TWR w, double (&b)[TCAP], int cnt)
{
    constexpr int Q = TW<G, TRI, FM>::Q;
    static_for<TCAP - 1>([&](auto jj) __attribute__((always_inline)) {
        constexpr int j = TCAP - 1 - jj;                    // 12 .. 1
        if (j < cnt) {
            double Lv[j];
            static_for<j>([&](auto i) __attribute__((always_inline)) { Lv[i] = w.sm[(kTL + tlidx(j, i)) * Q]; });
            static_for<j>([&](auto i) __attribute__((always_inline)) { b[i] = msub<FM>(b[i], b[j], Lv[i]); });
        }
    });
}
// constrained stationary point (auxiliary.c:314-354): lam* in b[] and in the LDS vector
TWT __device__ __forceinline__ void tcsp(TWR w, double (&b)[TCAP])
{
    constexpr int Q = TW<G, TRI, FM>::Q;
    tforward(w, w.reuse, TV_RHS);
    static_for<TCAP>([&](auto i) __attribute__((always_inline)) {
        const double z = w.sm[(kTV + TV_ZL + i) * Q];
        b[i] = (i < w.na) ? z : 0.0;
    });
    tbackward(w, b, w.na);
    w.reuse = w.na;
}
TWT __device__ __forceinline__ void tstore_lams(TWR w, const double (&b)[TCAP])
{
    constexpr int Q = TW<G, TRI, FM>::Q;
    double *ls = w.sm + TLAMS(w) * Q;
    if (w.sub == 0) static_for<TCAP>([&](auto i) __attribute__((always_inline)) { ls[i * Q] = b[i]; });
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
}
// auxiliary.c:357-376: the direction along which the singular system is unbounded, in b[] (and the lam* vector)
TWT __device__ __forceinline__ void tsingular_direction(TWR w, double (&b)[TCAP])
{
    constexpr int Q = TW<G, TRI, FM>::Q;
    const int s = w.sing;
    const double *Ls = w.sm + (kTL + tlidx(s, 0)) * Q;
    static_for<TCAP>([&](auto i) __attribute__((always_inline)) {
        const double l = (i < TCAP - 1) ? Ls[(i < TCAP - 1 ? i : 0) * Q] : 0.0;
        b[i] = (i < s) ? -l : 0.0;
    });
    tbackward(w, b, s);
    const bool flip = (tws_flag(w, s) & DAQP_LOWER) != 0;
    static_for<TCAP>([&](auto i) __attribute__((always_inline)) {
        const double v = (i == s) ? 1.0 : b[i];
        b[i] = (i <= s) ? (flip ? -v : v) : 0.0;
    });
}
// ratio test of auxiliary.c:277-311 (SOFT_WEIGHTS off) on lam* = b[]: returns the position to drop (or -1) after stepping
// lam towards lam*
TWT __device__ __forceinline__ int tblocking(TWR w, const double (&b)[TCAP])
{
    constexpr int Q = TW<G, TRI, FM>::Q;
    const int na = w.na;
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
    double lv[TCAP];
    static_for<TCAP>([&](auto i) __attribute__((always_inline)) { lv[i] = lm[i * Q]; });
    static_for<TCAP>([&](auto i) __attribute__((always_inline)) {
        if (blk & (1u << i)) {
            const double cand = regular ? -lv[i] / (b[i] - lv[i]) : -lv[i] / b[i];
            if (cand < alpha) { alpha = cand; rm = i; }
        }
    });
    if (rm < 0) return -1;
    if (w.sub == 0)
        static_for<TCAP>([&](auto i) __attribute__((always_inline)) {
            if (i < na) lm[i * Q] = regular ? lv[i] + alpha * (b[i] - lv[i]) : lv[i] + alpha * b[i];
        });
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    w.sing = kEmpty;
    return rm;
}
// u = -M_k' lam* in working-set order (auxiliary.c:46-88): lane s forms the components s, s+G, ... from the cached rows, then
// every lane of the group receives all of them
TWT __device__ __forceinline__ void tprimal(TWR w, const double (&b)[TCAP])
{
    constexpr int Q = TW<G, TRI, FM>::Q, CP = TNC / G;
    double uu[CP];
    static_for<CP>([&](auto t) __attribute__((always_inline)) { uu[t] = 0.0; });
    static_for<TCAP>([&](auto i) __attribute__((always_inline)) {
        if (i < w.na) {
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
    static_for<RPL / KG>([&](auto gg) __attribute__((always_inline)) {
        double mu[KG], du[KG], dl[KG], sc[KG];
        static_for<KG>([&](auto kk) __attribute__((always_inline)) {
            constexpr int k = KG * gg + kk;
            const int r = w.sub + G * k, rr = r < w.m ? r : 0;
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
            const bool open = r < w.m && !(sn & (DAQP_ACTIVE + DAQP_IMMUTABLE));
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
// into cache slot `slot`; sn = sense bits of the row.
TWT __device__ __forceinline__ double tappend(TWR w, int id, int slot, int sn)
{
    constexpr int Q = TW<G, TRI, FM>::Q, RPL = TW<G, TRI, FM>::RPL;
    const int na = w.na, n = w.n;
    {   // the owning lane stores its registers
        double *dst = w.sm + (kTR + slot * TNC) * Q;
        const bool owner = w.sub == (id & (G - 1));
        const int kq = id / G;
        static_for<RPL>([&](auto k) __attribute__((always_inline)) {
            if (owner && kq == k)
                static_for<TNC>([&](auto j) __attribute__((always_inline)) {
                    if constexpr (k < TRI && j < G * k) dst[j * Q] = 0.0; else dst[j * Q] = w.M[k][j];
                });
        });
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    }
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
    static_for<TCAP>([&](auto k) __attribute__((always_inline)) {
        g[k] = 0.0;
        if (k < na) {
            const int idk = (int)(k < 8 ? (w.id0 >> (6 * k)) & 63 : (w.id1 >> (6 * (k - 8))) & 63);
            const int j = (idk < w.ms) ? (c0 > idk ? c0 : idk) : c0;
            g[k] = dotrow((int)((w.slw >> (4 * k)) & 15), j);
        }
    });
    double dnew = dotrow(slot, c0);
    int ns_act = 0;
    if (w.has_soft) {
        static_for<TCAP>([&](auto k) __attribute__((always_inline)) { if (k < na && (tws_flag(w, k) & DAQP_SOFT)) ns_act++; });
        if (sn & DAQP_SOFT) { ns_act++; dnew += w.rho_soft; }
    }
    if (na == 0) return dnew;
    // l <- L \ g, column by column (every g_i receives its subtractions in ascending j: factorization.c:81-88); rows beyond
    // na carry unused values
    static_for<TCAP - 2>([&](auto j) __attribute__((always_inline)) {
        if (j < na - 1) {
            double Lv[TCAP - 1 - j];
            static_for<TCAP - 1 - j>([&](auto ii) __attribute__((always_inline)) { Lv[ii] = w.sm[(kTL + tlidx(j + 1 + ii, j)) * Q]; });
            static_for<TCAP - 1 - j>([&](auto ii) __attribute__((always_inline)) { g[j + 1 + ii] = msub<FM>(g[j + 1 + ii], Lv[ii], g[j]); });
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
TWT __device__ __forceinline__ void tpush(TWR w, int id, int sn, double lamv, double rhs)
{
    constexpr int Q = TW<G, TRI, FM>::Q;
    ttrace(w, id + 1);
    tsense_set(w, id, DAQP_ACTIVE, 0);
    const int slot = __ffs((int)~w.slotmask) - 1;
    w.slotmask |= 1u << slot;
    const double dnew = tappend(w, id, slot, sn | DAQP_ACTIVE);
    tws_set(w, w.na, id, slot, sn);
    if (w.sub == 0) {
        w.sm[(TLAM(w) + w.na) * Q] = lamv;
        w.sm[(kTV + TV_D + w.na) * Q] = dnew;
        w.sm[(kTV + TV_RHS + w.na) * Q] = rhs;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    w.na++;
}

// ---- LDL' row delete (factorization.c:112-151) + the bookkeeping of auxiliary.c:3-22; returns 1 if the factor became singular
TWT __device__ __forceinline__ int tdrop(TWR w, int r)
{
    constexpr int Q = TW<G, TRI, FM>::Q;
    const int na = w.na;
    const int idr = tws_id(w, r);
    ttrace(w, -(idr + 1));
    tsense_set(w, idr, 0, DAQP_ACTIVE);
    w.slotmask &= ~(1u << tws_slot(w, r));
    const int nupd = na - r - 1;
    double *lm = w.sm + TLAM(w) * Q;
    if (nupd > 0) {
        constexpr int WP = (TCAP - 1 + G - 1) / G;
        // column r below the diagonal: lane s keeps w_t for t = s, s+G, ...
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
        // Gill-Golub-Murray-Saunders C1 rank-one update of the trailing block, pivot by pivot
        double alpha = w.sm[(kTV + TV_D + r) * Q];
        static_for<TCAP - 1>([&](auto j) __attribute__((always_inline)) {
            if (j < nupd) {
                const double p = gbcast<G, j % G>(wv[j / G]);
                const double Di = w.sm[(kTV + TV_D + r + 1 + j) * Q];
                const double dbar = Di + alpha * p * p;
                // beta = p*alpha/dbar and alpha' = D_i*alpha/dbar: two divisions by the same number, one in the even and one
                // in the odd lanes
                const double num = (w.sub & 1) ? Di * alpha : p * alpha;
                const double quo = num / dbar;
                const double beta = gbcast<G, 0>(quo);
                alpha = gbcast<G, 1>(quo);
                if (w.sub == 0) w.sm[(kTV + TV_D + r + j) * Q] = dbar;
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

// one step of iterative refinement on the active rows (auxiliary.c:498-593); lam* in b[] on entry and on return
TWT __device__ __forceinline__ void trefine(TWR w)
{
    constexpr int Q = TW<G, TRI, FM>::Q;
    const int na = w.na;
    w.reuse = 0;
    double b[TCAP];
    {
        const double *ls = w.sm + TLAMS(w) * Q;
        static_for<TCAP>([&](auto i) __attribute__((always_inline)) { b[i] = ls[i * Q]; });
    }
    for (int i = 0; i < na; ++i) {
        const int id = tws_id(w, i), fl = tws_flag(w, i);
        const double *row = w.sm + (kTR + tws_slot(w, i) * TNC) * Q;
        const int j0 = id < w.ms ? id : 0;
        double mu = 0;
        static_for<TNC>([&](auto j) __attribute__((always_inline)) {
            const double t = mu + row[j * Q] * w.u[j];
            mu = (j >= j0 && j < w.n) ? t : mu;
        });
        double li = 0;
        static_for<TCAP>([&](auto k) __attribute__((always_inline)) { li = (k == i) ? b[k] : li; });
        double res = mu - (-w.sm[(kTV + TV_RHS + i) * Q]);      // d = -rhs exactly
        if (fl & DAQP_SOFT) res -= w.rho_soft * li;
        if (w.sub == 0) w.sm[(kTV + TV_XL + i) * Q] = res;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    tforward(w, 0, TV_XL);
    double dl[TCAP];
    static_for<TCAP>([&](auto i) __attribute__((always_inline)) {
        const double z = w.sm[(kTV + TV_ZL + i) * Q];
        dl[i] = (i < na) ? z : 0.0;
    });
    tbackward(w, dl, na);
    static_for<TCAP>([&](auto i) __attribute__((always_inline)) {
        if (i < na) { if (w.sub == 0) w.sm[(kTV + TV_XL + i) * Q] = dl[i]; b[i] += dl[i]; }
    });
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    for (int i = 0; i < na; ++i) {
        const int id = tws_id(w, i);
        const double *row = w.sm + (kTR + tws_slot(w, i) * TNC) * Q;
        const int j0 = id < w.ms ? id : 0;
        double di = 0;
        static_for<TCAP>([&](auto k) __attribute__((always_inline)) { di = (k == i) ? dl[k] : di; });
        static_for<TNC>([&](auto j) __attribute__((always_inline)) {
            const double t = w.u[j] - row[j * Q] * di;
            w.u[j] = (j >= j0 && j < w.n) ? t : w.u[j];
        });
    }
    double fv = w.soft;
    static_for<TNC>([&](auto j) __attribute__((always_inline)) { fv += w.u[j] * w.u[j]; });
    w.fval = fv;
    tstore_lams(w, b);
}

// bound (on the side its LOWER bit says) and sense bits of constraint id, from the owning lane's registers
TWT __device__ __forceinline__ void trow_lookup(TWR w, int id, double &bound, int &sn)
{
    const bool owner = w.sub == (id & (G - 1));
    const int s = (int)((w.rs >> (4 * (id / G))) & 15);
    sn = gor<G>(owner ? s : 0);
    bound = (sn & DAQP_LOWER) ? w.gdl[id] : w.gdu[id];
}

// ---------------------------------------------------------------------------------------------------------------------------
// daqp_ldp (daqp.c:6-108) + daqp_activate_constraints (auxiliary.c:399-479) + daqp_pivot_last (auxiliary.c:379-396) for the
// problems of a wave, in lockstep phases.  Per problem: st = what it does next; an edit request (add / drop) with its
// continuation; an activation cursor.  mode 1: only rebuild the working set from the ACTIVE bits.
// ---------------------------------------------------------------------------------------------------------------------------
enum : int { TST_ACT, TST_ITER, TST_EDIT, TST_DONE };
enum : int { TAFTER_NEXT_ITER, TAFTER_CYCLE_GUARD, TAFTER_ACT_POST };
enum : int { TACT_THEN_DONE, TACT_THEN_LOOP, TACT_THEN_NEXT_ITER, TACT_THEN_CYCLE_RESET };

struct TinyOut { int flag, iterations; };

TWT __device__ __forceinline__ TinyOut trun(TWR w, int mode, bool alive, bool need_activate, double *pend, unsigned long long t_start, double tick_s)
{
    constexpr int Q = TW<G, TRI, FM>::Q, RPL = TW<G, TRI, FM>::RPL;
    int flag = DAQP_EXIT_ITERLIMIT, it = 1, repaired = 0, stall = 0;
    double best = -1;
    const double fbound = 2 * w.stp->fval_bound;
    const int iter_limit = w.stp->iter_limit;
    const double progress_tol = w.stp->progress_tol;
    const int cycle_tol = w.stp->cycle_tol;
    const double time_limit = w.stp->time_limit;
    const bool tl_armed = time_limit > 0.0;
    int tl_skip = 0;
#define TTL_CHECK() (tl_armed && !tl_skip && (it & 31) == 0 && time_is_up(t_start, time_limit, tick_s))
    int depth = 0, req_add = 1, req_id = 0, req_r = 0, req_sn = 0, after = TAFTER_NEXT_ITER;
    double req_lam = 0, req_rhs = 0;
    int act_then = TACT_THEN_DONE, act_i = 0, act_next = 0, act_flag = 1;
    int st;
    if (!alive) st = TST_DONE;
    else if (mode == 1 || need_activate) {
        w.sing = kEmpty; w.na = 0; w.reuse = 0; w.slotmask = 0;
        act_then = (mode == 1) ? TACT_THEN_DONE : TACT_THEN_LOOP;
        st = TST_ACT;
    } else st = (it < iter_limit) ? TST_ITER : TST_DONE;

    while (__any(st != TST_DONE)) {
        // ---- activation cursor: the next ACTIVE-marked row in index order (auxiliary.c:399-479)
        if (st == TST_ACT) {
            int cand = kRowNone;
            static_for<RPL>([&](auto kk) __attribute__((always_inline)) {
                constexpr int k = RPL - 1 - kk;
                const int r = w.sub + G * k;
                if (r >= act_next && r < w.m && ((w.rs >> (4 * k)) & DAQP_ACTIVE)) cand = r;
            });
            cand = gmin_i<G>(cand);
            if (cand == kRowNone) {
                if (act_then == TACT_THEN_DONE) st = TST_DONE;
                else if (act_then == TACT_THEN_LOOP) {
                    if (act_flag < 0) { flag = act_flag; st = TST_DONE; }
                    else { it = 1; st = (it < iter_limit) ? TST_ITER : TST_DONE; }
                } else {
                    if (act_then == TACT_THEN_CYCLE_RESET) { stall = 0; best = -1; }
                    if (TTL_CHECK()) { flag = DAQP_EXIT_TIMELIMIT; st = TST_DONE; }
                    else { ++it; st = (it < iter_limit) ? TST_ITER : TST_DONE; }
                }
            } else {
                double bd; int sn;
                trow_lookup(w, cand, bd, sn);
                act_i = cand;
                req_add = 1; req_id = cand; req_sn = sn; req_lam = (sn & DAQP_LOWER) ? -1.0 : 1.0; req_rhs = -bd;
                depth = 0; after = TAFTER_ACT_POST; st = TST_EDIT;
            }
        }
        // ---- one iteration of daqp_ldp up to its working-set edit (daqp.c:12-64, 86-93)
        if (st == TST_ITER) {
            tl_skip = 0;
            const bool was_singular = (w.sing != kEmpty);
            double b[TCAP];
            if (!was_singular) tcsp(w, b); else { ttrace(w, kTraceSingular); tsingular_direction(w, b); }
            tstore_lams(w, b);
            const int blk = tblocking(w, b);
            if (blk >= 0) { req_add = 0; req_r = blk; depth = 0; after = TAFTER_NEXT_ITER; st = TST_EDIT; }
            else if (was_singular) { flag = DAQP_EXIT_INFEASIBLE; st = TST_DONE; }
            else {
                tprimal(w, b);
                int upper = 0, sn = 0, pick = kRowNone, pass = 0;
                double bd = 0;
                bool settled = false;
                for (;;) {      // (a second trip only after refine_active)
                    pick = tscan(w, upper, bd, sn, pass == 0);
                    if (pass == 0 && w.fval > fbound) { flag = DAQP_EXIT_INFEASIBLE; st = TST_DONE; settled = true; break; }
                    if (pick != kRowNone || pass == 1) break;
                    double dmin = DAQP_INF;
                    static_for<TCAP>([&](auto i) __attribute__((always_inline)) {
                        const double d = w.sm[(kTV + TV_D + i) * Q];
                        dmin = (i < w.na && d < dmin) ? d : dmin;
                    });
                    if (w.na > 2 && repaired != 1 && dmin < w.stp->refactor_tol) {     // daqp.c:33-46
                        repaired = 1; tl_skip = 1;
                        ttrace(w, kTraceRefactor);
                        const double *lm = w.sm + TLAM(w) * Q;
                        for (int i = 0; i < w.na; ++i) {
                            const int id = tws_id(w, i);
                            if (lm[i * Q] >= 0) tsense_set(w, id, 0, DAQP_LOWER); else tsense_set(w, id, DAQP_LOWER, 0);
                        }
                        w.sing = kEmpty; w.na = 0; w.reuse = 0; w.slotmask = 0;
                        act_then = TACT_THEN_NEXT_ITER; act_next = 0; act_flag = 1; st = TST_ACT;
                        settled = true;
                        break;
                    }
                    if (w.na > 0 && dmin < w.pivot_tol) {                                 // daqp.c:52-56
                        ttrace(w, kTraceRefine);
                        trefine(w);
                        pass = 1; tl_skip = 1;
                        continue;
                    }
                    break;
                }
                if (!settled) {
                    if (pick == kRowNone) {
                        flag = (w.soft > w.stp->primal_tol) ? DAQP_EXIT_SOFT_OPTIMAL : DAQP_EXIT_OPTIMAL;
                        st = TST_DONE;
                    } else {
                        // auxiliary.c:152-166: fix the side, lam <-> lam*, then add with multiplier +-1
                        if (upper) tsense_set(w, pick, 0, DAQP_LOWER); else tsense_set(w, pick, DAQP_LOWER, 0);
                        w.lamsw ^= 1;
                        req_add = 1; req_id = pick; req_sn = upper ? (sn & ~DAQP_LOWER) : (sn | DAQP_LOWER);
                        req_lam = upper ? 1.0 : -1.0; req_rhs = -bd;
                        depth = 0; after = pass ? TAFTER_NEXT_ITER : TAFTER_CYCLE_GUARD; st = TST_EDIT;
                    }
                }
            }
        }
        // ---- the edit: remove_constraint / add_constraint (auxiliary.c:3-44), one per problem and pass
        bool edited = false, sing_after_drop = false;
        if (st == TST_EDIT && !req_add) { sing_after_drop = tdrop(w, req_r) != 0; edited = true; }
        if (st == TST_EDIT && req_add && !edited) { tpush(w, req_id, req_sn, req_lam, req_rhs); edited = true; }
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
                    if (w.sub == 0) {
                        pend[3 * depth] = __hiloint2double(tws_flag(w, r), tws_id(w, r));
                        pend[3 * depth + 1] = w.sm[(TLAM(w) + r) * Q];
                        pend[3 * depth + 2] = w.sm[(kTV + TV_RHS + r) * Q];
                    }
                    depth++;
                    req_add = 0; req_r = r; more = true;
                } else if (depth > 0 && w.sing == kEmpty) {
                    depth--;
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    const double key = gpick<G>(w.sub == 0 ? pend[3 * depth] : 0.0, w.sub == 0);
                    req_id = __double2loint(key); req_sn = __double2hiint(key);
                    req_lam = gpick<G>(w.sub == 0 ? pend[3 * depth + 1] : 0.0, w.sub == 0);
                    req_rhs = gpick<G>(w.sub == 0 ? pend[3 * depth + 2] : 0.0, w.sub == 0);
                    req_add = 1; more = true;
                }
            }
            if (!more) {
                bool next_iter = false;
                if (after == TAFTER_ACT_POST) {
                    if (w.sing == kEmpty) { act_next = act_i + 1; st = TST_ACT; }
                    else {
                        const int last = tws_id(w, w.na - 1), lastflag = tws_flag(w, w.na - 1);
                        if (lastflag & DAQP_IMMUTABLE) {   // a new equality depends on the active ones: consistent => ignore it
                            double b[TCAP];
                            tsingular_direction(w, b);
                            tstore_lams(w, b);
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
                            if (resid <= w.stp->primal_tol * scale && resid >= -w.stp->primal_tol * scale) act_next = act_i + 1;
                            else { act_flag = DAQP_EXIT_OVERDETERMINED_INITIAL; act_next = kRowNone; }
                        } else {
                            int bad = 0;
                            static_for<RPL>([&](auto k) __attribute__((always_inline)) {   // rows >= act_i: unactivated equalities are an error, the rest are cleaned
                                const int rr = w.sub + G * k;
                                const int sn = (int)((w.rs >> (4 * k)) & 15);
                                const bool later = rr >= act_i && rr < w.m && (sn & DAQP_ACTIVE);
                                if (later && (sn & DAQP_IMMUTABLE)) bad = 1;
                                if (later && !(sn & DAQP_IMMUTABLE)) w.rs &= ~((unsigned long long)DAQP_ACTIVE << (4 * k));
                            });
                            bad = gor<G>(bad);
                            w.slotmask &= ~(1u << tws_slot(w, w.na - 1));
                            w.na--;
                            w.sing = kEmpty;
                            act_flag = bad ? DAQP_EXIT_OVERDETERMINED_INITIAL : 1;
                            act_next = kRowNone;
                        }
                        st = TST_ACT;
                    }
                } else if (after == TAFTER_CYCLE_GUARD) {   // daqp.c:66-85
                    next_iter = true;
                    if (w.fval - best < progress_tol) {
                        if (stall++ > cycle_tol) {
                            if (repaired == 1) { flag = DAQP_EXIT_CYCLE; st = TST_DONE; next_iter = false; }
                            else {
                                repaired = 1;
                                ttrace(w, kTraceCycleReset);
                                w.sing = kEmpty; w.na = 0; w.reuse = 0; w.slotmask = 0;
                                act_then = TACT_THEN_CYCLE_RESET; act_next = 0; act_flag = 1; st = TST_ACT;
                                next_iter = false;
                            }
                        }
                    } else { best = w.fval; stall = 0; }
                } else next_iter = true;
                if (next_iter) {
                    if (TTL_CHECK()) { flag = DAQP_EXIT_TIMELIMIT; st = TST_DONE; }
                    else { ++it; st = (it < iter_limit) ? TST_ITER : TST_DONE; }
                }
            }
        }
    }
#undef TTL_CHECK
    TinyOut o;
    o.flag = (mode == 1) ? act_flag : flag;
    o.iterations = it;
    return o;
}

#undef TWT
#undef TWR
} // namespace daqp_amd
