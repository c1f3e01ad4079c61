// wave_ldp_reg.hip.h -- register-centric variant of the one-wavefront LDP iteration, for
// working sets of at most 64 rows (n + n_soft + 1 <= 64).  Same algorithm and the same
// floating-point operation order as wave_ldp.hip.h (reference src/daqp.c, src/auxiliary.c,
// src/factorization.c); what changes is where the state lives:
//
//   lane i  <-> position i of the working set : constraint id, row-cache slot, sense flags,
//               lam, lam*, D_i, x_i/z_i of the LDL' solve, right-hand side  -- all VGPRs
//   lane r  <-> constraint rows r, r+64, ...  : the rows of M themselves (NB x NP double2),
//               d_upper, d_lower, -primal_tol*scaling, sense                -- all VGPRs/AGPRs
//   LDS                                         : packed L, the active-row cache (slot-indexed,
//               so a removal moves no row), u
//
// One wave per SIMD is all the LDS budget allows, so nothing hides an LDS round trip: every
// substitution loop therefore preloads its L entries eight steps at a time BEFORE entering the
// dependency chain, and cross-lane traffic is v_readlane / DPP, never LDS.
#pragma once
#include <utility>
#include <type_traits>
#include "wave_ldp.hip.h"

namespace daqp_amd {

constexpr int kChunk = 8;

// compile-time loop: f(integral_constant<int, 0>) ... f(integral_constant<int, N-1>).  Register
// arrays are only ever indexed through these, so an index is a constant by construction (a
// `#pragma unroll` that the optimizer declines turns the whole array into scratch memory).
template <class F, int... Is>
__device__ __forceinline__ void static_for_impl(F &&f, std::integer_sequence<int, Is...>) { (f(std::integral_constant<int, Is>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void static_for(F &&f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }

// the register shape whose problems may outgrow the 64 lanes (n = 64: up to 65 working-set rows) hands them over; compiled out of every other shape
// (and so does every kernel that keeps an fp32 IMAGE of M in its registers -- IMG = 1, see RWave -- whose LDS holds fewer working-set rows than the
// problem may reach: those problems go to the IMG = 0 kernel of the same shape)
// IMG = 2: as IMG = 1 with the LAST row block (at most 32 rows: 128 < m <= 160 at NB = 3) split over two lanes per row -- NP / 2 image registers
// for it instead of NP: C2's 150 rows are 126 registers instead of 150, which is what lets the rest of the state stay out of scratch
template <int NB, int IMG> constexpr int kImgFullBlocks = (IMG == 2) ? NB - 1 : NB;
// rows of the row view (d_upper, d_lower, -primal_tol scaling in LDS, one after the other): 64 per row block; IMG = 2: the last block has 32
template <int NB, int IMG> constexpr int kRowvStride = (IMG == 2) ? 64 * NB - 32 : 64 * NB;
template <int NB, int NP, int IMG = 0> constexpr bool kRegHandOver = (NB == 2 && NP == 32) || IMG != 0;
constexpr int kRegHandOverFlag = -1000;   // rrun's verdict "not mine" (never reaches the caller: no reference exit flag is near it)
// IMG = 0: the rows of M themselves live in the registers (fp64: 4 NB NP registers, the whole file at C2's shape -> one wave per SIMD).
// IMG = 1 (default arithmetic only): the registers hold an fp32 IMAGE of M (2 NB NP registers) that SCREENS the feasibility scan; a verdict is
// taken from it only when it is provably the fp64 scan's (rscan_rows), a row that enters the working set is fetched in fp64 from the blocked
// image in HBM / L2 (rfetch_row), and the kernel fits two waves per SIMD (docs/NEXT_two_waves_per_simd.md).
template <int NB, int NP, bool FM, int IMG = 0>
struct RWave {
    typedef double gv2d_ __attribute__((ext_vector_type(2)));
    typedef typename std::conditional<IMG != 0, float, double>::type mreg;
    const DAQP_GLOBAL(gv2d_) *msrc;    // IMG = 1: this problem's blocked fp64 image [nblk][npair][64][2]
    float *u32;                        // IMG = 1: u rounded to fp32 (LDS), the screening scan's operand
    int npair;
    // IMG != 0: the active-row cache is TIERED.  Slots < cache_slots are rows in LDS (rowc); the others live in this problem's global scratch
    // (rowg, slot s at (s - cache_slots) ldr: L2-resident, 400 contiguous bytes a row at C2's shape), and LDS row `cache_slots` is where a row
    // bound for such a slot is staged while it is appended.  New rows take the lowest free slot, so the scratch is touched only by working sets
    // beyond cache_slots rows.  cache_slots >= max_rows: everything in LDS.
    int cache_slots;
    DAQP_GLOBAL(double) *rowg;
    int n, m, ms, ldr;
    double *L, *rowc, *u, *pend_lam;   // LDS
    int *pend_id;                      // LDS
    // working-set view (lane i = position i)
    int wsid, slot, wflag;
    double lam, lams, D, xl, zl, drhs;
    // row view (lane r + 64*bb = constraint row).  At one wave per SIMD the kernel owns the
    // whole unified 512-entry register file; the compiler parks what exceeds the 256
    // architectural VGPRs in AGPRs (v_accvgpr_read on use).
    mreg Mx[NB][NP], My[NB][NP];   // M[row][2t], M[row][2t+1]
    // d_upper, d_lower and -primal_tol*scaling of every row live in LDS (rowv[r], rowv[R + r], rowv[2R + r], R = 64*NB): they are read once per iteration at the end of the scan, and 18 more registers held across the whole
    // loop push the allocator into scratch spills (the register file is full: 300 registers of M + the working set)
    double *rowv;
    typedef typename std::conditional<(NB > 4), unsigned long long, unsigned>::type rs_t;     // (more than four row blocks: the image-only shapes)
    rs_t rs;       // sense bits of this lane's rows, 8 bits per row block (a register, never an array)
    // uniform
    int na, reuse, sing, has_soft;
    int max_rows;   // (2,32) only -- see kRegHandOver: an add that finds this many rows in place hands the problem over
    int hi_slot;                        // highest row-cache slot ever used (the top of the cache doubles as a prefetch buffer)
    unsigned long long slotmask;
    double fval, soft;
    const DAQP_GLOBAL(DAQPSettings) *stp;   // device copy of the settings (cold fields)
    double dual_tol, sing_tol, pivot_tol, rho_soft;   // hot tolerances, loaded once
    DAQP_GLOBAL(int) *trace; int trace_cap, trace_len;
    long long *prof;                    // LDS, 8 phase counters (NULL: off)
};

// The lane number as a value the optimizer cannot see through: LDS addresses of the form "base + 8*lane" are loop
// invariant, so LICM hoists them out of the state-machine loop, the register file (full: M alone is 300 registers)
// cannot hold them, and they come back as scratch reloads -- a trip to memory to save one v_lshl_add.  Addresses
// built from lane_now() are recomputed where they are used.
__device__ __forceinline__ int lane_now() { int l = lane_id(); asm volatile("" : "+v"(l)); return l; }

// Cycle-counter probes (per-state / per-phase, tools/gpu_profile.py).  They are compiled in by default and switched on
// per batch at run time (daqp_batch_enable_profile): measured on gfx950, the build WITH the (disabled) probes is the
// faster one -- k_ldp_reg 7.53 ms vs 7.95 ms per 20 k QPs of config C2 -- because the probe branches split the giant
// basic blocks of the state machine and the register allocator, working on a completely full register file, does
// better on the pieces.  -DDAQP_AMD_NO_PROFILE removes them.
#ifndef DAQP_AMD_NO_PROFILE
constexpr bool kProfile = true;
#else
constexpr bool kProfile = false;
#endif
#define RPROF_T0(w) long long prof_t0_ = (kProfile && (w).prof) ? (long long)__builtin_readcyclecounter() : 0
#define RPROF_ACC(w, slot) do { if (kProfile && (w).prof) { const long long t1_ = (long long)__builtin_readcyclecounter(); if (lane_id() == 0) (w).prof[slot] += t1_ - prof_t0_; prof_t0_ = t1_; } } while (0)

__device__ __forceinline__ int rli(int v, int src) { return __builtin_amdgcn_readlane(v, src); }
__device__ __forceinline__ float rlf32(float v, int src) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src)); }

// x - a*b and x + a*b.  FM = false: two roundings, the reference's arithmetic (the file is compiled with -ffp-contract=off).
// FM = true (default arithmetic mode of the library, where M = A R^-1 already comes from the matrix cores): one v_fma_f64,
// i.e. one rounding -- results differ from the reference's in the last bits, decisions are compared at the north_star bar
// (tests/test_gpu_fast_mode.py); one instruction less per term of every chain and of the feasibility scan.
template <bool FM> __device__ __forceinline__ double msub(double x, double a, double b) { if constexpr (FM) return __builtin_fma(-a, b, x); else return x - a * b; }
template <bool FM> __device__ __forceinline__ double madd(double x, double a, double b) { if constexpr (FM) return __builtin_fma(a, b, x); else return x + a * b; }

template <int NB, int NP, bool FM, int IMG>
__device__ __forceinline__ void rtrace(RWave<NB, NP, FM, IMG> &w, int ev)
{
    if (w.trace) {
        if (lane_id() == 0 && w.trace_len < w.trace_cap) w.trace[w.trace_len] = ev;
        w.trace_len++;
    }
}

// --- row-view accessors ----------------------------------------------------------------------
// sense words are < 256 (bits ACTIVE..SLACK_FIXED): block bb of this lane sits in byte bb of w.rs
template <int NB, int NP, bool FM, int IMG>
__device__ __forceinline__ int rsense_get(const RWave<NB, NP, FM, IMG> &w, int bb) { return (int)((w.rs >> (8 * bb)) & 0xffu); }
template <int NB, int NP, bool FM, int IMG>
__device__ __forceinline__ int sense_of(const RWave<NB, NP, FM, IMG> &w, int id)   // id wave-uniform
{
    return rli(rsense_get(w, id >> 6), id & 63);
}
template <int NB, int NP, bool FM, int IMG>
__device__ __forceinline__ void sense_set(RWave<NB, NP, FM, IMG> &w, int id, int set_bits, int clear_bits)
{
    if (lane_id() == (id & 63)) {
        const int sh = 8 * (id >> 6);
        typedef typename RWave<NB, NP, FM, IMG>::rs_t rs_t;
        w.rs = (w.rs | ((rs_t)set_bits << sh)) & ~((rs_t)clear_bits << sh);
    }
}
// bound of constraint id: broadcast every block's candidate first, THEN pick (selecting between
// array elements before the readlane gets folded into a dynamic index => scratch)
template <int NB, int NP, bool FM, int IMG>
__device__ __forceinline__ double bound_of(const RWave<NB, NP, FM, IMG> &w, int id, bool lower)
{
    return w.rowv[(lower ? kRowvStride<NB, IMG> : 0) + id];     // wave-uniform address: an LDS broadcast
}

// rowc[slot] <- row id: the owning lane stores its registers, NP unconditional 16-byte writes (the row
// stride is >= 2*NP and rows are 16-byte aligned; entries beyond n are the zero padding of M)
template <int NB, int NP, bool FM, int IMG>
__device__ __forceinline__ void rfetch_row(RWave<NB, NP, FM, IMG> &w, int id, int slot)
{
    const int lane = lane_id();
    if constexpr (IMG != 0) {
        // the exact row comes from the blocked image: pair t of row id sits at [id / 64][t][id % 64] -- lane t fetches its 16 bytes straight
        // into the cache slot (global_load_lds: LDS address = slot base + 16 lane); one trip to L2 / HBM per added constraint, which the
        // SIMD's other wave covers
        typedef typename RWave<NB, NP, FM, IMG>::gv2d_ gv2d_;
        const DAQP_GLOBAL(gv2d_) *src = w.msrc + ((size_t)(id >> 6) * w.npair) * 64 + (id & 63) + (size_t)lane * 64;
        if (lane < w.npair)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                             (__attribute__((address_space(3))) void *)(w.rowc + (size_t)slot * w.ldr), 16, 0, 0);
        copy_wait();
        WSYNC();
        return;
    }
    double2 *dst = reinterpret_cast<double2 *>(w.rowc + (size_t)slot * w.ldr);
    static_for<NB>([&](auto bb) __attribute__((always_inline)) {
        if ((id >> 6) == bb && lane == (id & 63)) {
            static_for<NP>([&](auto t) __attribute__((always_inline)) {
                double2 v; v.x = w.Mx[bb][t]; v.y = w.My[bb][t];
                dst[t] = v;
            });
        }
    });
    WSYNC();
}

// factorization.c:4-15 with the loads of each group of 8 issued before its arithmetic
template <bool FM>
__device__ __forceinline__ double dot4_pipelined(const double *a, const double *b, int len)
{
    double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
    int i = 0;
    for (; i + 7 < len; i += 8) {
        double x[8], y[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) { x[q] = a[i + q]; y[q] = b[i + q]; }
        s0 = madd<FM>(s0, x[0], y[0]); s1 = madd<FM>(s1, x[1], y[1]); s2 = madd<FM>(s2, x[2], y[2]); s3 = madd<FM>(s3, x[3], y[3]);
        s0 = madd<FM>(s0, x[4], y[4]); s1 = madd<FM>(s1, x[5], y[5]); s2 = madd<FM>(s2, x[6], y[6]); s3 = madd<FM>(s3, x[7], y[7]);
    }
    for (; i + 3 < len; i += 4) {
        double x[4], y[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) { x[q] = a[i + q]; y[q] = b[i + q]; }
        s0 += x[0] * y[0]; s1 += x[1] * y[1]; s2 += x[2] * y[2]; s3 += x[3] * y[3];
    }
    for (; i < len; i++) s0 += a[i] * b[i];
    return (s0 + s1) + (s2 + s3);
}

// Measured on gfx950 at one wave per SIMD (tools/ubench.hip): a ROLLED loop step costs ~67 cycles
// against ~23 unrolled (taken branch ~40), a uniform branch ~18, any LDS instruction 25-40 cycles of
// issue time.  Hence: chains run in straight-line groups of 8 with the loads of the group issued
// first, and steps beyond the end are padded with exact zeros (x - 0*y == x) instead of guarded.

// Lane numbers of the chains below are COMPILE-TIME constants (static chunks of 8 lanes, a chunk skipped by one
// uniform branch once past the end): v_readlane with an immediate lane costs ~8 cycles, with a computed lane ~15
// (s_add + SGPR-index hazard, tools/ubench2.hip), and a constant column turns every L address into
// "lane base + immediate offset".

// The chains below run over eight static chunks of eight lanes, each behind its own "is this chunk live" branch (~18 cycles
// whether taken or not).  Working sets of up to 16 rows pay for the upper six chunks with ONE branch, those of up to 32 rows for the upper four.
// The smallest register shape (n <= 16: working sets of a handful of rows) runs chunks of FOUR steps instead: at three waves per
// SIMD that kernel is bound by instruction issue, and a chain over 3 rows then executes 4 steps, not 8 (C3: -9 % instructions).
#ifndef DAQP_AMD_CHAIN4_UP_TO
#define DAQP_AMD_CHAIN4_UP_TO 8     // register shapes with NP up to this run chunks of four (measured: C2's shape is slower with them)
#endif
template <int NP> constexpr int chain_g() { return NP <= DAQP_AMD_CHAIN4_UP_TO ? 4 : 8; }
template <int G = 8, class F> __device__ __forceinline__ void chunks_up(int live, F &&f)
{
    if constexpr (G == 8) {
        static_for<2>([&](auto c) __attribute__((always_inline)) { f(c); });
        if (live > 16) {
            static_for<2>([&](auto c) __attribute__((always_inline)) { f(std::integral_constant<int, c + 2>{}); });
            if (live > 32) static_for<4>([&](auto c) __attribute__((always_inline)) { f(std::integral_constant<int, c + 4>{}); });
        }
    } else {
        static_for<2>([&](auto c) __attribute__((always_inline)) { f(c); });
        if (live > 8) {
            static_for<2>([&](auto c) __attribute__((always_inline)) { f(std::integral_constant<int, c + 2>{}); });
            if (live > 16) {
                static_for<4>([&](auto c) __attribute__((always_inline)) { f(std::integral_constant<int, c + 4>{}); });
                if (live > 32) static_for<8>([&](auto c) __attribute__((always_inline)) { f(std::integral_constant<int, c + 8>{}); });
            }
        }
    }
}
template <int G = 8, class F> __device__ __forceinline__ void chunks_down(int live, F &&f)   // f(cc), chunk 64/G - 1 - cc: top chunks first
{
    if constexpr (G == 8) {
        if (live > 16) {
            if (live > 32) static_for<4>([&](auto cc) __attribute__((always_inline)) { f(cc); });
            static_for<2>([&](auto cc) __attribute__((always_inline)) { f(std::integral_constant<int, cc + 4>{}); });
        }
        static_for<2>([&](auto cc) __attribute__((always_inline)) { f(std::integral_constant<int, cc + 6>{}); });
    } else {
        if (live > 8) {
            if (live > 16) {
                if (live > 32) static_for<8>([&](auto cc) __attribute__((always_inline)) { f(cc); });
                static_for<4>([&](auto cc) __attribute__((always_inline)) { f(std::integral_constant<int, cc + 8>{}); });
            }
            static_for<2>([&](auto cc) __attribute__((always_inline)) { f(std::integral_constant<int, cc + 12>{}); });
        }
        static_for<2>([&](auto cc) __attribute__((always_inline)) { f(std::integral_constant<int, cc + 14>{}); });
    }
}
// b <- L' \ b over the leading cnt positions (column-oriented; product order b_j * L[j][i]).
// Lanes >= cnt must hold b == 0.
template <int NB, int NP, bool FM, int IMG>
__device__ __forceinline__ double rbackward(RWave<NB, NP, FM, IMG> &w, double b, int cnt)
{
    const int lane = lane_id();
    const double *Ll = w.L + lane_now();
    // step j updates lanes with lane < j < cnt  <=>  (unsigned)(j - 1 - lane) < (unsigned)(cnt - 1 - lane) for lane < cnt:
    // one add and one compare per step on per-lane registers, no wave-uniform mask per step (those end up as
    // spilled SGPR pairs).  Lanes >= cnt may pick up unused values; nothing reads them.
    const unsigned room = (unsigned)(cnt - 1 - lane);
    const int nlane = -1 - lane;
    constexpr int G = chain_g<NP>();
    chunks_down<G>(cnt, [&](auto cc) __attribute__((always_inline)) {
        constexpr int c = 64 / G - 1 - cc;
        if (G * c < cnt && cnt > 1) {
            double Lb[G];
            static_for<G>([&](auto q) __attribute__((always_inline)) {   // unconditional loads (any address inside the LDS allocation is fine)
                constexpr int j = G * c + G - 1 - q;
                Lb[q] = Ll[tri(j)];
            });
            static_for<G>([&](auto q) __attribute__((always_inline)) {
                constexpr int j = G * c + G - 1 - q;
                if constexpr (j >= 1) {
                    const double bj = rl(b, j);
                    const double t = msub<FM>(b, bj, Lb[q]);
                    b = ((unsigned)(j + nlane) < room) ? t : b;
                }
            });
        }
    });
    return b;
}
// sum over the wave by tree (default arithmetic mode only: not the reference's order): 4 DPP steps inside each row of 16, then 4 readlanes
__device__ __forceinline__ double wave_sum(double v)
{
    v += dpp_f64<0xB1>(v);
    v += dpp_f64<0x4E>(v);
    v += dpp_f64<0x141>(v);
    v += dpp_f64<0x140>(v);
    return (rl(v, 0) + rl(v, 16)) + (rl(v, 32) + rl(v, 48));
}
// inclusive prefix sum over the lanes (default arithmetic mode only): DPP row_shr steps inside each row of 16, then the three row
// totals by readlane
template <int CTRL>
__device__ __forceinline__ double dpp_f64_or0(double v)       // lanes without a source receive +0.0
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xF, 0xF, false);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
}
template <int CTRL>
__device__ __forceinline__ double dpp_f64_old(double old, double v)       // lanes without a source receive `old`
{
    const int lo = __builtin_amdgcn_update_dpp(__double2loint(old), __double2loint(v), CTRL, 0xF, 0xF, false);
    const int hi = __builtin_amdgcn_update_dpp(__double2hiint(old), __double2hiint(v), CTRL, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
}
// inclusive scan of the affine maps s -> a_j s + b_j, composed in lane order (lane j: the map of lanes 0..j): the linear recurrence
// s_{j+1} = a_j s_j + b_j for all j at once.  rows: 16-lane rows that hold anything but identities (1, 0).
__device__ __forceinline__ void wave_scan_affine(double &a, double &b, int rows)
{
#define DAQP_AFF_STEP(CTRL) { const double as_ = dpp_f64_old<CTRL>(1.0, a), bs_ = dpp_f64_old<CTRL>(0.0, b); b = __builtin_fma(a, bs_, b); a = a * as_; }
    DAQP_AFF_STEP(0x111) DAQP_AFF_STEP(0x112) DAQP_AFF_STEP(0x114) DAQP_AFF_STEP(0x118)
#undef DAQP_AFF_STEP
    if (rows > 1) {
        const int row = lane_id() >> 4;
#pragma unroll
        for (int k = 1; k < 4; ++k) {
            if (k < rows) {
                const double at = rl(a, 16 * k - 1), bt = rl(b, 16 * k - 1);
                const bool in = row == k;
                b = __builtin_fma(a, in ? bt : 0.0, b);
                a = a * (in ? at : 1.0);
            }
        }
    }
}
__device__ __forceinline__ double wave_scan_incl(double v)
{
    v += dpp_f64_or0<0x111>(v);   // row_shr:1
    v += dpp_f64_or0<0x112>(v);   // row_shr:2
    v += dpp_f64_or0<0x114>(v);   // row_shr:4
    v += dpp_f64_or0<0x118>(v);   // row_shr:8
    const int lane = lane_id();
    const double s0 = rl(v, 15), s1 = rl(v, 31), s2 = rl(v, 47);
    double carry = (lane >= 16) ? s0 : 0.0;
    carry += (lane >= 32) ? s1 : 0.0;
    carry += (lane >= 48) ? s2 : 0.0;
    return v + carry;
}
// ordered sum: acc - p_0 - p_1 - ... - p_{cnt-1} (p must be 0 in lanes >= cnt)
template <int G = 8>
__device__ __forceinline__ double ordered_sub(double acc, double p, int cnt)
{
    chunks_up<G>(cnt, [&](auto c) __attribute__((always_inline)) {
        if (G * c < cnt) static_for<G>([&](auto q) __attribute__((always_inline)) { acc -= rl(p, G * c + q); });
    });
    return acc;
}
// x_i = rhs_i - sum_{j<i} L[i][j] x_j for rows i >= from (j ascending), column-oriented: lane <-> row, the
// lane's own L entries for 8 columns preloaded, x_j broadcast by v_readlane once final.
// In: x = final values for lanes < from; rhs for lanes in [from, na); 0 beyond.
template <int NB, int NP, bool FM, int IMG>
__device__ __forceinline__ double rforward(RWave<NB, NP, FM, IMG> &w, double x, double rhs, int from)
{
    const int lane = lane_id(), na = w.na;
    const bool pending = lane >= from && lane < na;
    x = pending ? rhs : (lane < na ? x : 0.0);
    if (from >= na) return x;          // nothing open (x was carried through a removal)
    if (from == na - 1 && na > 1) {
        // the usual case after an add: only the last row is open.  Its products in parallel, then the j-ordered
        // chain of subtractions on broadcast operands -- the same operations as the sweep below for that row
        const double p = (lane < na - 1) ? w.L[tri(na - 1) + lane_now()] * x : 0.0;
        const double last = FM ? rl(rhs, na - 1) - wave_sum(p) : ordered_sub<chain_g<NP>()>(rl(rhs, na - 1), p, na - 1);   // (default mode: a tree instead of the k-ordered chain)
        return (lane == na - 1) ? last : x;
    }
    const int pl = pending ? lane : -1;
    const double *Lr = w.L + tri(lane_now());
    constexpr int G = chain_g<NP>();
    chunks_up<G>(na - 1, [&](auto c) __attribute__((always_inline)) {
        if (G * c < na - 1) {
            double Lk[G];
            static_for<G>([&](auto q) __attribute__((always_inline)) { Lk[q] = Lr[G * c + q]; });
            static_for<G>([&](auto q) __attribute__((always_inline)) {
                constexpr int j = G * c + q;
                const double xj = rl(x, j);
                const double t = msub<FM>(x, Lk[q], xj);
                x = (pl > j) ? t : x;     // rows > j that are still open; steps j >= na-1 select nothing (pl < na)
            });
        }
    });
    return x;
}

// factorization.c:4-15 for all rows of the working set at once when no simple bounds are involved: TWO lanes per row
// (lane 2k+h: row k, h = 0 carries the reference's partial sums (s0,s1), h = 1 carries (s2,s3)), so a lane reads
// every other 16-byte pair of its row and of the new row: half the LDS instructions and a quarter of the VALU work
// of one lane per row.  Element e < 4*(n/4) goes to chain e%4 in ascending order, the n%4 tail elements go to s0 one
// after the other, and the result is (s0+s1)+(s2+s3) -- exactly the reference's dot_row.  Row na is the new row itself.
// rows of the scratch tier fetched per trip to L2 (IMG != 0; rprimal_u, rdots_two_lanes)
#ifndef DAQP_IMG_TIER2
#define DAQP_IMG_TIER2 8
#endif
constexpr int kTier2 = DAQP_IMG_TIER2;
template <int NB, int NP, bool FM, int IMG>
__device__ __forceinline__ double rdots_two_lanes(RWave<NB, NP, FM, IMG> &w, int newslot, const double *Mi)
{
    const int lane = lane_id(), na = w.na, n = w.n, nq = n >> 2, h = lane & 1;
    const double2 *rb = reinterpret_cast<const double2 *>(Mi) + h;
    double g = 0;
    for (int p0 = 0; p0 <= na; p0 += 32) {
        const int k = p0 + (lane >> 1);
        const int sl = __shfl(w.slot, k & 63);
        const int sk = (k < na && (IMG == 0 || sl < w.cache_slots)) ? sl : newslot;   // rows beyond na (and rows outside the LDS tier: done below): the new row again (finite, unused)
        const double *rowk = w.rowc + (size_t)sk * w.ldr;
        const double2 *ra = reinterpret_cast<const double2 *>(rowk) + h;
        double sa = 0, sb = 0;
        static_for<(NP / 2 + 3) / 4>([&](auto G) __attribute__((always_inline)) {
            if (4 * G + 3 < nq) {
                double2 x[4], y[4];
                static_for<4>([&](auto q) __attribute__((always_inline)) { x[q] = ra[2 * (4 * G + q)]; y[q] = rb[2 * (4 * G + q)]; });
                static_for<4>([&](auto q) __attribute__((always_inline)) { sa = madd<FM>(sa, x[q].x, y[q].x); sb = madd<FM>(sb, x[q].y, y[q].y); });
            } else if (4 * G < nq) {
                static_for<4>([&](auto q) __attribute__((always_inline)) {
                    if (4 * G + q < nq) {
                        const double2 x = ra[2 * (4 * G + q)], y = rb[2 * (4 * G + q)];
                        sa += x.x * y.x; sb += x.y * y.y;
                    }
                });
            }
        });
        for (int e = 4 * nq; e < n; ++e) {                       // tail: all of it on s0
            const double t = sa + rowk[e] * Mi[e];
            sa = (h == 0) ? t : sa;
        }
        const double part = sa + sb;
        const double tot = part + __shfl_xor(part, 1);           // (s0+s1)+(s2+s3), the same bits in both lanes
        const double gk = __shfl(tot, (2 * (lane - p0)) & 63);
        if (lane >= p0 && lane < p0 + 32) g = gk;
    }
    if constexpr (IMG != 0) {
        // rows of the scratch tier: lane <-> component, four rows in flight, each dot by tree (default arithmetic)
        unsigned long long m2 = __ballot(lane < na && w.slot >= w.cache_slots);
        const double mine = (lane < n) ? Mi[lane_now()] : 0.0;
        while (m2) {
            int ii[kTier2];
            double rr[kTier2];
            static_for<kTier2>([&](auto c) __attribute__((always_inline)) {
                ii[c] = m2 ? __ffsll((long long)m2) - 1 : -1;
                if (m2) m2 &= m2 - 1;
                const int so = (ii[c] >= 0) ? (rli(w.slot, ii[c] & 63) - w.cache_slots) * w.ldr : 0;
                rr[c] = (lane < n && ii[c] >= 0) ? w.rowg[so + lane] : 0.0;      // (no row: exactly zero -- the scratch is not initialised)
            });
            static_for<kTier2>([&](auto c) __attribute__((always_inline)) {
                if (ii[c] >= 0) {               // (wave-uniform)
                    const double sum = wave_sum(rr[c] * mine);
                    if (lane == ii[c]) g = sum;
                }
            });
        }
    }
    return g;
}

// ---------------------------------------------------------------------------------------
// LDL' row append (factorization.c:21-111); returns the new pivot D[na]
// ---------------------------------------------------------------------------------------
template <int NB, int NP, bool FM, int IMG>
__device__ __forceinline__ double rldl_append(RWave<NB, NP, FM, IMG> &w, int id, int newslot, int sn_id)
{
    const int lane = lane_id(), na = w.na, n = w.n, base = tri(na);
    const int gslot = newslot;                                  // the slot the row is known by
    if constexpr (IMG != 0) newslot = newslot < w.cache_slots ? newslot : w.cache_slots;    // ... and where it is in LDS during the append (the staging row)
    rfetch_row(w, id, newslot);
    const int c0 = id < w.ms ? id : 0;
    w.sing = kEmpty;
    const double *Mi = w.rowc + (size_t)newslot * w.ldr;
    if constexpr (IMG != 0) {       // a row of the scratch tier: its copy goes out now (nothing waits for the store)
        if (gslot >= w.cache_slots && lane < n) w.rowg[(gslot - w.cache_slots) * w.ldr + lane] = Mi[lane_now()];
    }
    double g = 0;
    if (lane <= na) {
        const int idk = (lane < na) ? w.wsid : id;
        const int sk = (lane < na && (IMG == 0 || w.slot < w.cache_slots)) ? w.slot : newslot;     // (rows of the scratch tier: below)
        const int j = (lane < na && idk < w.ms) ? (c0 > idk ? c0 : idk) : c0;
        if (w.ms != 0) g = dot4_pipelined<FM>(w.rowc + (size_t)sk * w.ldr + j, Mi + j, n - j);
    }
    if constexpr (IMG != 0) {
        if (w.ms != 0) {
            // simple bounds with rows in the scratch tier: their dots lane <-> component from the start column on (factorization.c:64-72), by tree
            unsigned long long m2 = __ballot(lane < na && w.slot >= w.cache_slots);
            const double mine = (lane < n) ? Mi[lane_now()] : 0.0;
            while (m2) {
                int ii[kTier2];
                double rr[kTier2];
                static_for<kTier2>([&](auto c) __attribute__((always_inline)) {
                    ii[c] = m2 ? __ffsll((long long)m2) - 1 : -1;
                    if (m2) m2 &= m2 - 1;
                    const int so = (ii[c] >= 0) ? (rli(w.slot, ii[c] & 63) - w.cache_slots) * w.ldr : 0;
                    rr[c] = (lane < n && ii[c] >= 0) ? w.rowg[so + lane] : 0.0;
                });
                static_for<kTier2>([&](auto c) __attribute__((always_inline)) {
                    if (ii[c] >= 0) {               // (wave-uniform)
                        const int idk = rli(w.wsid, ii[c] & 63);
                        const int j0 = (idk < w.ms) ? (c0 > idk ? c0 : idk) : c0;
                        const double sum = wave_sum(lane >= j0 ? rr[c] * mine : 0.0);
                        if (lane == ii[c]) g = sum;
                    }
                });
            }
        }
    }
    if (w.ms == 0) g = rdots_two_lanes(w, newslot, Mi);
    int ns_act = 0;
    if (w.has_soft) ns_act = __popcll(__ballot(lane < na && (w.wflag & DAQP_SOFT))) + ((sn_id & DAQP_SOFT) ? 1 : 0);
    double dnew = rl(g, na);
    if (sn_id & DAQP_SOFT) dnew += w.rho_soft;
    if (na == 0) return dnew;
    // forward substitution with L, column by column; each lane preloads its own row of L 8 columns ahead
    {
        const int pl = lane < na ? lane : -1;
        const double *Lr = w.L + tri(lane_now());
        constexpr int G = chain_g<NP>();
        chunks_up<G>(na - 1, [&](auto c) __attribute__((always_inline)) {
            if (G * c < na - 1) {
                double Lk[G];
                static_for<G>([&](auto q) __attribute__((always_inline)) { Lk[q] = Lr[G * c + q]; });
                static_for<G>([&](auto q) __attribute__((always_inline)) {
                    constexpr int j = G * c + q;
                    const double lj = rl(g, j);
                    const double t = msub<FM>(g, Lk[q], lj);
                    g = (pl > j) ? t : g;
                });
            }
        });
    }
    double p = 0;
    if (lane < na) {
        const double t = g;
        const double lk = t / w.D;
        w.L[base + lane_now()] = lk;
        p = t * lk;
    }
    double acc = FM ? dnew - wave_sum(p) : ordered_sub<chain_g<NP>()>(dnew, p, na);
    if (acc < w.sing_tol || na >= n + ns_act) { w.sing = na; acc = 0; }
    WSYNC();
    return acc;
}

// ---------------------------------------------------------------------------------------
// LDL' row delete (factorization.c:112-151)
// ---------------------------------------------------------------------------------------
template <int NB, int NP, bool FM, int IMG>
__device__ __forceinline__ void rldl_delete(RWave<NB, NP, FM, IMG> &w, int r)
{
    const int lane = lane_id(), na = w.na;
    if (na == r + 1) return;
    const int nupd = na - r - 1;
    RPROF_T0(w);
    double wv = (lane < nupd) ? w.L[tri(r + 1 + lane_now()) + r] : 0.0;
    // move rows r+1.. up by one and drop column r, element-parallel over the packed destination range
    // [tri(r), tri(na-1)).  Destination e always reads from a higher address, so ascending chunks with
    // "read all, then write all" never clobber a live source.  Row of e: float sqrt + one branch-free
    // correction each way (|error| < 1 for e < 2^21).
    {
        const int e0 = tri(r), e1 = tri(na - 1);
        constexpr int U = 2;
        for (int cb = e0; cb < e1; cb += 64 * U) {
            double tmp[U];
#pragma unroll
            for (int q = 0; q < U; ++q) {
                const int e = cb + q * 64 + lane;
                int i = (int)((__builtin_sqrtf(8.0f * (float)e + 1.0f) - 1.0f) * 0.5f);
                i += (tri(i + 1) <= e) ? 1 : 0;
                i -= (tri(i) > e) ? 1 : 0;
                const int j = e - tri(i);
                tmp[q] = w.L[(e < e1) ? tri(i + 1) + j + (j >= r ? 1 : 0) : 0];
            }
            WSYNC();
#pragma unroll
            for (int q = 0; q < U; ++q) {
                const int e = cb + q * 64 + lane;
                if (e < e1) w.L[e] = tmp[q];
            }
            WSYNC();
        }
    }
    // Gill-Golub-Murray-Saunders C1 update: lane t <-> trailing row r+t (new numbering); its L
    // entries for 8 consecutive columns are read before, and written back after, the chain
    RPROF_ACC(w, 24);
    double alpha = rl(w.D, r);
    double Dn = w.D;
    const double Drot = __shfl(w.D, (lane + r + 1) & 63);     // D_{r+1+j} in lane j: compile-time lane numbers below
    // Default arithmetic: the CSP's x = L^-1 rhs rides through the update (x~_2 = K^-1 (x_2 + p x_r), the same recurrence the rows
    // of L go through, on one more scalar per pivot), so that the next CSP has no rows to re-solve after a removal -- provided x was
    // complete (the removal follows a CSP).  The reference re-solves rows >= r; same mathematics, other rounding.
    const double Xrot = FM ? __shfl(w.xl, (lane + r + 1) & 63) : 0.0;
    const double xr = FM ? rl(w.xl, r) : 0.0;
    double sx = 0, Xn = w.xl;
    const int pl = lane < nupd ? lane : -1;
    double *Lr = w.L + tri(r + lane_now()) + r;
    // (the smallest register shape keeps the chain: its updates are a handful of pivots, where the scans' fixed cost does not
    //  pay -- C3 1.90 ms with the chain, 1.96 ms with the two passes)
    if constexpr (FM && NP > 8) {
        // Default arithmetic: the same update with no scalar work per pivot.  With t = 1/alpha the recurrence alpha' = alpha D / dbar
        // becomes the running sum t' = t + p^2 / D, and the p_j themselves do not depend on alpha or beta at all (w_i -= p_j L_ij
        // uses the OLD column): pass A is that chain alone -- afterwards lane j holds p_j --, then ONE prefix sum and four
        // lane-parallel divisions serve every pivot at once (alpha_j = 1 / t_j, dbar_j = D_j + alpha_j p_j^2, beta_j = p_j alpha_j /
        // dbar_j), the CSP's x rides along as ONE scan of affine maps, and pass B applies w_i -= p_j L_ij; L_ij += beta_j w_i with
        // both scalars ready: two broadcasts and two fused multiply-adds per pivot, no guard inside.
        // (Measured on C2: the pass B below alone, with the x recurrence and the guards of the chain still inside, cost what the whole
        // chain had cost -- ~180 cycles per pivot are ~25 issued instructions, not division latency; with x as a scan and no guards
        // the removal's update went from 6.6 k to 5.2 k cycles.  Keeping BOTH forms in the kernel cost more than either: 31.2 ms.)
        // A singular factor has its zero pivot in the LAST position only: its 1/D = inf reaches t of lanes that nothing reads, and
        // beta, dbar are formed from dbar itself (beta = p alpha / dbar, dbar = D + alpha p^2: finite there, as in the chain).
        {
            double wa = wv;
            chunks_up(nupd, [&](auto c) __attribute__((always_inline)) {
                if (8 * c < nupd) {
                    double La[8];
                    static_for<8>([&](auto q) __attribute__((always_inline)) {
                        constexpr int j = 8 * c + q;
                        La[q] = (pl > j) ? Lr[j] : 0.0;
                    });
                    static_for<8>([&](auto q) __attribute__((always_inline)) {
                        constexpr int j = 8 * c + q;
                        wa = __builtin_fma(-rl(wa, j), La[q], wa);      // (lanes <= j and lanes beyond the range: La == 0)
                    });
                }
            });
            const double rD = 1.0 / ((lane < nupd) ? Drot : 1.0);
            const double pp = (lane < nupd) ? wa * wa : 0.0;
            const double tv = 1.0 / alpha + wave_scan_incl(pp * rD);     // lane j: t_{j+1}
            const double rtv = 1.0 / tv;
            double rtp;                                                    // lane j: alpha_j = 1 / t_j (lane 0: alpha itself)
            {
                const int lo = __builtin_amdgcn_update_dpp(__double2loint(alpha), __double2loint(rtv), 0x138, 0xF, 0xF, false);   // wave_shr:1
                const int hi = __builtin_amdgcn_update_dpp(__double2hiint(alpha), __double2hiint(rtv), 0x138, 0xF, 0xF, false);
                rtp = __hiloint2double(hi, lo);
            }
            double betav;
            {   // the new pivots, back in working-set numbering (before pass B: nothing of this stays live across it)
                const double dbarv = __builtin_fma(pp, rtp, Drot);
                betav = (lane < nupd) ? wa * rtp / dbarv : 0.0;
                const double dsh = __shfl(dbarv, (lane - r) & 63);
                if (lane >= r && lane < r + nupd) Dn = dsh;
            }
            {   // the CSP's x through the update: s_{j+1} = s_j + beta_j (X_j + p_j (x_r - s_j)) is the affine recurrence
                // s -> (1 - beta_j p_j) s + beta_j (X_j + p_j x_r): one scan instead of three dependent operations per pivot
                const bool in = lane < nupd;
                double a = in ? __builtin_fma(-betav, wa, 1.0) : 1.0;
                double bb = in ? betav * __builtin_fma(wa, xr, Xrot) : 0.0;
                wave_scan_affine(a, bb, (nupd + 15) >> 4);
                const double sj = dpp_f64_old<0x138>(0.0, bb);                       // wave_shr:1 -- lane j: s_j
                const double xtv = __builtin_fma(wa, xr - sj, Xrot);
                const double xsh = __shfl(xtv, (lane - r) & 63);
                if (lane >= r && lane < r + nupd) Xn = xsh;
            }
            chunks_up(nupd, [&](auto c) __attribute__((always_inline)) {
                if (8 * c < nupd) {
                    double Lc[8];
                    static_for<8>([&](auto q) __attribute__((always_inline)) {
                        constexpr int j = 8 * c + q;
                        Lc[q] = (pl > j) ? Lr[j] : 0.0;
                    });
                    static_for<8>([&](auto q) __attribute__((always_inline)) {   // (no guards inside: lanes <= j and pivots beyond the
                        constexpr int j = 8 * c + q;                             //  range see zeros -- p, beta, their L entries)
                        wv = __builtin_fma(-rl(wa, j), Lc[q], wv);
                        Lc[q] = __builtin_fma(rl(betav, j), wv, Lc[q]);
                    });
                    static_for<8>([&](auto q) __attribute__((always_inline)) {
                        constexpr int j = 8 * c + q;
                        if (pl > j) Lr[j] = Lc[q];
                    });
                }
            });
        }
    } else {
    constexpr int G = chain_g<NP>();
    chunks_up<G>(nupd, [&](auto c) __attribute__((always_inline)) {
        if (G * c < nupd) {
            double Lc[G];
            static_for<G>([&](auto q) __attribute__((always_inline)) {
                constexpr int j = G * c + q;
                Lc[q] = (pl > j) ? Lr[j] : 0.0;
            });
            static_for<G>([&](auto q) __attribute__((always_inline)) {
                constexpr int j = G * c + q;
                if (j < nupd) {
                    const double p = rl(wv, j);
                    const double Di = rl(Drot, j);
                    const double dbar = Di + alpha * p * p;
                    const double beta = p * alpha / dbar;
                    alpha = Di * alpha / dbar;
                    if (lane == r + j) Dn = dbar;
                    if constexpr (FM) {
                        const double xt = __builtin_fma(-p, sx, __builtin_fma(p, xr, rl(Xrot, j)));
                        sx = __builtin_fma(beta, xt, sx);
                        if (lane == r + j) Xn = xt;
                    }
                    if (pl > j) {
                        wv = msub<FM>(wv, p, Lc[q]);
                        Lc[q] = madd<FM>(Lc[q], beta, wv);
                    }
                }
            });
            static_for<G>([&](auto q) __attribute__((always_inline)) {
                constexpr int j = G * c + q;
                if (pl > j) Lr[j] = Lc[q];
            });
        }
    });
    }
    w.D = Dn;
    if constexpr (FM) {
        if (w.reuse >= na) {          // x was complete: keep it so (rdrop_core sets reuse to the new na)
            w.xl = Xn;
            if (lane >= r && lane < na - 1) w.zl = Xn / Dn;
            w.reuse = na + 64;        // marker: "carried"
        }
    }
    WSYNC();
    RPROF_ACC(w, 25);
}

// lane i <- lane i+1 for lanes >= r (closing the gap a removed working-set position leaves): one DPP move per
// dword (wave_shl:1, whole-wave shift on GFX9-family CDNA) instead of a ds_bpermute round trip through the LDS
__device__ __forceinline__ int shl1(int v) { return __builtin_amdgcn_update_dpp(v, v, 0x130, 0xF, 0xF, false); }
__device__ __forceinline__ int shift_from(int v, int r) { const int up = shl1(v); return lane_id() >= r ? up : v; }
__device__ __forceinline__ double shift_from(double v, int r)
{
    const int lo = shl1(__double2loint(v)), hi = shl1(__double2hiint(v));
    return lane_id() >= r ? __hiloint2double(hi, lo) : v;
}

template <int NB, int NP, bool FM, int IMG>
__device__ __forceinline__ int rdrop_core(RWave<NB, NP, FM, IMG> &w, int r) // auxiliary.c:3-22
{
    const int lane = lane_id();
    const int idr = rli(w.wsid, r);
    rtrace(w, -(idr + 1));
    sense_set(w, idr, 0, DAQP_ACTIVE);
    w.slotmask &= ~(1ull << rli(w.slot, r));
    rldl_delete(w, r);
    w.na--;
    w.wsid = shift_from(w.wsid, r);
    w.slot = shift_from(w.slot, r);
    w.wflag = shift_from(w.wflag, r);
    w.lam = shift_from(w.lam, r);
    w.drhs = shift_from(w.drhs, r);
    if (w.reuse > w.na + 32) w.reuse = w.na;       // (x carried through the removal: nothing to re-solve)
    else if (r < w.reuse) w.reuse = r;
    if (w.na > 0 && rl(w.D, w.na - 1) < w.sing_tol) {
        w.sing = w.na - 1;
        if (lane == w.na - 1) w.D = 0;
        return 1;
    }
    return 0;
}

template <int NB, int NP, bool FM, int IMG>
__device__ __forceinline__ void rpush_core(RWave<NB, NP, FM, IMG> &w, int id, double lamv) // auxiliary.c:27-40
{
    const int lane = lane_id();
    rtrace(w, id + 1);
    sense_set(w, id, DAQP_ACTIVE, 0);
    const int sn = sense_of(w, id);
    const int newslot = __ffsll((long long)~w.slotmask) - 1;
    w.slotmask |= 1ull << newslot;
    if (newslot > w.hi_slot) w.hi_slot = newslot;
    const double dnew = rldl_append(w, id, newslot, sn);
    const bool lower = (sn & DAQP_LOWER) != 0;
    const double bd = bound_of(w, id, lower);
    if (lane == w.na) {
        w.wsid = id; w.slot = newslot; w.wflag = sn; w.lam = lamv; w.D = dnew; w.drhs = -bd;
    }
    w.na++;
}

// ---------------------------------------------------------------------------------------
// per-iteration kernels
// ---------------------------------------------------------------------------------------
template <int NB, int NP, bool FM, int IMG>
__device__ __forceinline__ void rsolve_csp(RWave<NB, NP, FM, IMG> &w) // auxiliary.c:314-354
{
    const int lane = lane_id(), na = w.na, from = w.reuse;
    long long tq = (kProfile && w.prof) ? (long long)__builtin_readcyclecounter() : 0;
    w.xl = rforward(w, w.xl, w.drhs, from);
    if (lane >= from && lane < na) w.zl = w.xl / w.D;
    if (kProfile && w.prof) { const long long t1 = (long long)__builtin_readcyclecounter(); if (lane == 0) w.prof[11] += t1 - tq; tq = t1; }
    const double b = rbackward(w, (lane < na) ? w.zl : 0.0, na);
    if (kProfile && w.prof) { const long long t1 = (long long)__builtin_readcyclecounter(); if (lane == 0) w.prof[12] += t1 - tq; tq = t1; }
    if (lane < na) w.lams = b;
    w.reuse = na;
}

template <int NB, int NP, bool FM, int IMG>
__device__ __forceinline__ void rsingular_direction(RWave<NB, NP, FM, IMG> &w) // auxiliary.c:357-376
{
    const int lane = lane_id(), s = w.sing;
    double b = (lane < s) ? -w.L[tri(s) + lane] : 0.0;
    b = rbackward(w, b, s);
    const bool flip = (rli(w.wflag, s) & DAQP_LOWER) != 0;
    if (lane <= s) {
        const double v = (lane == s) ? 1.0 : b;
        w.lams = flip ? -v : v;
    }
}

// ratio test of auxiliary.c:277-311 (SOFT_WEIGHTS off): returns the position to drop (or kBig)
// after stepping lam towards lam*; the removal itself is the caller's single DROP site
template <int NB, int NP, bool FM, int IMG>
__device__ __forceinline__ int rblocking_test(RWave<NB, NP, FM, IMG> &w)
{
    const int lane = lane_id(), na = w.na;
    const double dtol = w.dual_tol;
    const bool regular = (w.sing == kEmpty);
    double bv = DAQP_INF;
    int bi = kBig, aux = 0;
    bool blocking = false;
    if (lane < na) {
        blocking = !(w.wflag & DAQP_IMMUTABLE);
        if (w.wflag & DAQP_LOWER) { if (w.lams < dtol) blocking = false; }
        else if (w.lams > -dtol) blocking = false;
    }
    if (!__any(blocking)) return kBig;      // the usual case (two iterations out of three): no division, no wave-wide minimum
    if (blocking) {
        const double cand = regular ? -w.lam / (w.lams - w.lam) : -w.lam / w.lams;
        if (cand < bv) { bv = cand; bi = lane; }
    }
    wave_argmin(bv, bi, aux);
    if (bi == kBig) return kBig;
    const double alpha = bv;
    if (lane < na) w.lam = regular ? w.lam + alpha * (w.lams - w.lam) : w.lam + alpha * w.lams;
    w.sing = kEmpty;
    return bi;
}

// u = -M_k' lam*  (auxiliary.c:46-88): lane <-> component j, working-set order, rows preloaded 8 ahead
template <int NB, int NP, bool FM, int IMG>
__device__ __forceinline__ void rprimal_u(RWave<NB, NP, FM, IMG> &w)
{
    const int lane = lane_id(), na = w.na, n = w.n;
    double uu = 0;
    // per working-set position (in its lane): element offset of its cached row (0 beyond na: a finite row) and
    // its multiplier (0 beyond na: the padding steps subtract 0 * finite, exact)
    const bool in_lds = lane < na && (IMG == 0 || w.slot < w.cache_slots);
    const int soff = in_lds ? w.slot * w.ldr : 0;
    const double lz = in_lds ? w.lams : 0.0;
    const double *rc = w.rowc + lane_now();
    constexpr int G = chain_g<NP>();
    chunks_up<G>(na, [&](auto c) __attribute__((always_inline)) {
        if (G * c < na) {
            double rv[G], li[G];
            static_for<G>([&](auto q) __attribute__((always_inline)) {
                constexpr int i = G * c + q;
                li[q] = rl(lz, i);
                rv[q] = rc[rli(soff, i)];
            });
            static_for<G>([&](auto q) __attribute__((always_inline)) { uu = msub<FM>(uu, rv[q], li[q]); });
        }
    });
    if constexpr (IMG != 0) {
        // rows of the scratch tier (working sets beyond cache_slots rows): four loads in flight, lane <-> component
        unsigned long long m2 = __ballot(lane < na && w.slot >= w.cache_slots);
        while (m2) {
            double rr[kTier2], ll[kTier2];
            static_for<kTier2>([&](auto c) __attribute__((always_inline)) {
                const bool has = m2 != 0;
                const int i = has ? __ffsll((long long)m2) - 1 : 0;
                ll[c] = has ? rl(w.lams, i) : 0.0;
                const int so = has ? (rli(w.slot, i) - w.cache_slots) * w.ldr : 0;
                if (has) m2 &= m2 - 1;
                rr[c] = (lane < n && has) ? w.rowg[so + lane] : 0.0;             // (no row: exactly zero -- the scratch is not initialised)
            });
            static_for<kTier2>([&](auto c) __attribute__((always_inline)) { uu = msub<FM>(uu, rr[c], ll[c]); });
        }
    }
    WSYNC();
    if (lane < n) w.u[lane_now()] = uu;
    if constexpr (IMG != 0) w.u32[lane_now()] = (lane < n) ? (float)uu : 0.0f;   // the screening scan's operand (64 floats: zero from n on)
    double fv = 0;
    if (w.has_soft) {
        const double sq = (lane < na && (w.wflag & DAQP_SOFT)) ? w.lams * w.lams : 0.0;
        const unsigned long long sm = __ballot(lane < na && (w.wflag & DAQP_SOFT));
        for (int i = 0; i < na; ++i) if ((sm >> i) & 1) fv += rl(sq, i);
    }
    fv = fv * w.rho_soft;
    w.soft = fv;
    // IMG = 1: |u|^2 here, by tree, from the lanes that hold u (the fp64 scan forms it on the way; the screening scan never sees u in fp64)
    if constexpr (IMG != 0) w.fval = fv + wave_sum(lane < n ? uu * uu : 0.0);
    WSYNC();
}

// feasibility scan + most-violated pick (auxiliary.c:89-198), everything but u from registers
// the value is materialised in a VGPR at this point of the program (an optimisation barrier for that value only)
__device__ __forceinline__ void pin_vgpr(double &x) { asm volatile("" : "+v"(x)); }

template <int NB, int NP, bool FM, int IMG>
__device__ __forceinline__ int rscan_rows(RWave<NB, NP, FM, IMG> &w, int &upper, bool with_fval)
{
    const int lane = lane_id();
    double bv = 0.0;
    int bi = kBig, bup = 0;
    double fv = w.soft;
    const double2 *u2 = reinterpret_cast<const double2 *>(w.u);
    double mu[NB];
    static_for<NB>([&](auto bb) __attribute__((always_inline)) { mu[bb] = 0; });
    // NB independent k-ordered chains per lane; u broadcast from LDS four pairs at a time, the next group's loads
    // issued before this group's arithmetic (two 16-register buffers: a whole-row preload would push that many
    // more rows of M out to the AGPRs, and every AGPR operand costs two v_accvgpr_read per scan).  Zero padding
    // (pairs beyond n/2) adds +0.0 and leaves every sum unchanged.
    constexpr int NG = (NP + 3) / 4;
    double2 ub[2][4];
    static_for<4>([&](auto h) __attribute__((always_inline)) { if constexpr (h < NP) ub[0][h] = u2[h]; });
    static_for<NG>([&](auto g) __attribute__((always_inline)) {
        if constexpr (g + 1 < NG)
            static_for<4>([&](auto h) __attribute__((always_inline)) { if constexpr (4 * (g + 1) + h < NP) ub[(g + 1) & 1][h] = u2[4 * (g + 1) + h]; });
        __builtin_amdgcn_sched_barrier(0);   // the next group's loads stay ahead of this group's arithmetic
        static_for<4>([&](auto h) __attribute__((always_inline)) {
            constexpr int t = 4 * g + h;
            if constexpr (t < NP) {
                const double ux = ub[g & 1][h].x, uy = ub[g & 1][h].y;
                static_for<NB>([&](auto bb) __attribute__((always_inline)) {
                    mu[bb] = madd<FM>(mu[bb], w.Mx[bb][t], ux);
                    mu[bb] = madd<FM>(mu[bb], w.My[bb][t], uy);
                });
                if (with_fval) { fv = madd<FM>(fv, ux, ux); fv = madd<FM>(fv, uy, uy); }   // j-ordered |u|^2 (auxiliary.c:85-86)
            }
        });
        // pin this group's partial sums here: without it the optimizer sinks whole chains below the loop (towards
        // their only use), which keeps every u pair alive at once
        static_for<NB>([&](auto bb) __attribute__((always_inline)) { pin_vgpr(mu[bb]); });
        pin_vgpr(fv);
    });
    double du_[NB], dl_[NB], bn_[NB];
    static_for<NB>([&](auto bb) __attribute__((always_inline)) {
        const int r = bb * 64 + lane_now();
        du_[bb] = w.rowv[r]; dl_[bb] = w.rowv[kRowvStride<NB, IMG> + r]; bn_[bb] = w.rowv[2 * kRowvStride<NB, IMG> + r];
    });
    // selection without branches (a branch lets the compiler sink a whole block's chain into it, serialising the blocks)
    static_for<NB>([&](auto bb) __attribute__((always_inline)) {
        const int r = bb * 64 + lane;
        const bool open = r < w.m && !(rsense_get(w, bb) & (DAQP_ACTIVE + DAQP_IMMUTABLE));
        const double cu = du_[bb] - mu[bb], cl = mu[bb] - dl_[bb];
        const bool up = open && cu < bv && cu < bn_[bb];
        const bool lo = open && !up && cl < bv && cl < bn_[bb];
        bv = up ? cu : (lo ? cl : bv);
        bi = (up || lo) ? r : bi;
        bup = up ? 1 : (lo ? 0 : bup);
    });
    if (with_fval) w.fval = fv;
    wave_argmin(bv, bi, bup);
    upper = bup;
    return bi;
}

// IMG = 1: the same pick from the fp32 image in the registers -- when that is provably what the fp64 scan picks.  The rule is the workgroup
// kernel's (wg_ldp.hip.h wg_scan32 / wscan): with E >= |fl32(M_r . u) - M_r . u| for every row (unit rows, fma chains of NP terms, two
// rounded operands per term, one final addition: E = (NP + 8) 2^-24 |u|), the most violated row is accepted only if it leads the
// runner-up by more than 2E, violates its own threshold by more than E and its side is unambiguous by more than 2E; "nothing violated"
// only if every open row clears its threshold by more than E; anything else -- and any non-finite sum -- is decided by the fp64 pass
// over the blocked image (rscan_rows_stream).  tools/band_stats.py: 0.04 % of C2's scans end there.  Expects w.fval = soft + |u|^2.
template <int NB, int NP, bool FM, int IMG>
__device__ __forceinline__ int rscan_rows_stream(RWave<NB, NP, FM, IMG> &w, int &upper)
{
    // the fp64 scan with the rows streamed from the blocked image: block by block, G pairs in flight per lane
    const int lane = lane_id();
    double bv = 0.0;
    int bi = kBig, bup = 0;
    const double2 *u2 = reinterpret_cast<const double2 *>(w.u);
    const int nblk = (w.m + 63) >> 6, np = w.npair;
    for (int bb = 0; bb < nblk; ++bb) {
        const int r = bb * 64 + lane;
        const int lr = (r < w.m) ? lane : 0;
        typedef typename RWave<NB, NP, FM, IMG>::gv2d_ gv2d_;
        const DAQP_GLOBAL(gv2d_) *src = w.msrc + ((size_t)bb * np) * 64 + lr;
        double mu = 0;
        constexpr int G = 5;
        for (int t0 = 0; t0 < np; t0 += G) {
            gv2d_ mm[G];
#pragma unroll
            for (int g = 0; g < G; ++g) { const int t = (t0 + g < np) ? t0 + g : np - 1; mm[g] = src[(size_t)t * 64]; }
#pragma unroll
            for (int g = 0; g < G; ++g) if (t0 + g < np) { const double2 uk = u2[t0 + g]; mu = __builtin_fma(mm[g].x, uk.x, mu); mu = __builtin_fma(mm[g].y, uk.y, mu); }
        }
        const int sn = (int)((w.rs >> (8 * bb)) & 0xffu);
        const bool open = r < w.m && !(sn & (DAQP_ACTIVE + DAQP_IMMUTABLE));
        const int rr = (r < w.m) ? r : 0;
        const double du = w.rowv[rr], dl = w.rowv[kRowvStride<NB, IMG> + rr], bn = w.rowv[2 * kRowvStride<NB, IMG> + rr];
        const double cu = du - mu, cl = mu - dl;
        const bool up = open && cu < bv && cu < bn;
        const bool lo = open && !up && cl < bv && cl < bn;
        bv = up ? cu : (lo ? cl : bv);
        bi = (up || lo) ? r : bi;
        bup = up ? 1 : (lo ? 0 : bup);
    }
    wave_argmin(bv, bi, bup);
    upper = bup;
    return bi;
}
template <int NB, int NP, bool FM, int IMG>
__device__ __forceinline__ int rscan_rows_img(RWave<NB, NP, FM, IMG> &w, int &upper)
{
    static_assert(IMG != 0 && FM, "the screening image belongs to the default arithmetic");
    const int lane = lane_id();
    typedef float f2 __attribute__((ext_vector_type(2)));
    const f2 *u2 = reinterpret_cast<const f2 *>(w.u32);
    float ax[NB], ay[NB];
    static_for<NB>([&](auto bb) __attribute__((always_inline)) { ax[bb] = 0.0f; ay[bb] = 0.0f; });
    constexpr int NBF = kImgFullBlocks<NB, IMG>, NPH = (NP + 1) / 2, GU = 5;
    // u pairs five at a time (broadcast reads), then their multiply-adds: a whole-row preload would take as many registers again as a block of the image
    static_for<(NP + GU - 1) / GU>([&](auto g) __attribute__((always_inline)) {
        f2 uk[GU];
        static_for<GU>([&](auto k) __attribute__((always_inline)) { if constexpr (GU * g + k < NP) uk[k] = u2[GU * g + k]; });
        __builtin_amdgcn_sched_barrier(0);
        static_for<GU>([&](auto k) __attribute__((always_inline)) {
            constexpr int t = GU * g + k;
            if constexpr (t < NP)
                static_for<NBF>([&](auto bb) __attribute__((always_inline)) {
                    ax[bb] = __builtin_fmaf(w.Mx[bb][t], uk[k].x, ax[bb]);
                    ay[bb] = __builtin_fmaf(w.My[bb][t], uk[k].y, ay[bb]);
                });
        });
        __builtin_amdgcn_sched_barrier(0);
    });
    if constexpr (IMG == 2) {
        // the last row block holds at most 32 rows: TWO lanes per row (lane 2k + h: row 64 (NB-1) + k, pairs NPH h .. NPH h + NPH - 1; pairs beyond
        // the row meet u = 0), half the registers; the halves meet by one DPP swap and move to the row's own lane (lane k) by one permute
        const f2 *uh = u2 + NPH * (lane & 1);
        static_for<(NPH + GU - 1) / GU>([&](auto g) __attribute__((always_inline)) {
            f2 uk[GU];
            static_for<GU>([&](auto k) __attribute__((always_inline)) { if constexpr (GU * g + k < NPH) uk[k] = uh[GU * g + k]; });
            __builtin_amdgcn_sched_barrier(0);
            static_for<GU>([&](auto k) __attribute__((always_inline)) {
                constexpr int t = GU * g + k;
                if constexpr (t < NPH) {
                    ax[NB - 1] = __builtin_fmaf(w.Mx[NB - 1][t], uk[k].x, ax[NB - 1]);
                    ay[NB - 1] = __builtin_fmaf(w.My[NB - 1][t], uk[k].y, ay[NB - 1]);
                }
            });
            __builtin_amdgcn_sched_barrier(0);
        });
        const float part = ax[NB - 1] + ay[NB - 1];
        const float both = part + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(part), 0xB1, 0xF, 0xF, false));
        ax[NB - 1] = __shfl(both, (2 * lane) & 63);
        ay[NB - 1] = 0.0f;
    }
    // E, rounded UP in single precision (one v_sqrt_f32 instead of a double-precision square root's dozen instructions)
    const double E = (double)(1.0002f * (float)(NP + 8) * 5.9604644775390625e-08f * __builtin_sqrtf((float)(w.fval - w.soft) * 1.0000003f) + 1e-37f);
    // Per lane only what the verdict's FIRST question needs: the smallest slack s = min(d_upper - mu, mu - d_lower) of its open rows, the runner-up
    // and the row.  Everything else -- the winner's threshold, side and side gap -- is formed once, for the winner, on wave-uniform values.
    double s1 = DAQP_INF, s2 = DAQP_INF;
    int i1 = kBig, bad = 0;
    float msum[NB];
    static_for<NB>([&](auto bb) __attribute__((always_inline)) {
        const int r = bb * 64 + lane;
        const int rr = bb * 64 + lane_now();
        const bool open = r < w.m && !(rsense_get(w, bb) & (DAQP_ACTIVE + DAQP_IMMUTABLE));
        const double du = w.rowv[rr], dl = w.rowv[kRowvStride<NB, IMG> + rr];
        msum[bb] = ax[bb] + ay[bb];
        const double mu = (double)msum[bb];
        if (open && !(msum[bb] - msum[bb] == 0.0f)) bad = 1;
        const double sm = __builtin_fmin(du - mu, mu - dl);
        const double s = open ? sm : (double)DAQP_INF;
        const bool first = s < s1;
        s2 = first ? s1 : __builtin_fmin(s, s2);
        i1 = first ? r : i1;
        s1 = __builtin_fmin(s, s1);
    });
    double bv = s1;
    int bi = i1, aux = lane;
    wave_argmin(bv, bi, aux);                       // aux: the lane that holds the winner
    const double other = (lane == aux && bi != kBig) ? s2 : s1;
    const double w2 = wave_min(other);
    const bool anybad = __ballot(bad) != 0;
    if (!anybad) {
        // every open row's exact slack is >= bv - E: with bv >= E none is below zero, let alone below its (negative) threshold
        if (bv >= E) { upper = 0; return kBig; }                                                   // certainly nothing violated
        if (bi != kBig && bv + 2.0 * E < w2) {                                                      // THE smallest slack, for certain
            const int wb = bi >> 6;
            float mw = msum[0];
            static_for<NB>([&](auto bb) __attribute__((always_inline)) { if (bb > 0 && wb == bb) mw = msum[bb]; });
            const double mu1 = (double)rlf32(mw, aux);
            const double du1 = w.rowv[bi], dl1 = w.rowv[kRowvStride<NB, IMG> + bi], bn1 = w.rowv[2 * kRowvStride<NB, IMG> + bi];   // (wave-uniform addresses)
            const double cu1 = du1 - mu1, cl1 = mu1 - dl1;
            const bool up = cu1 <= cl1;
            const double gap = up ? cl1 - cu1 : cu1 - cl1;
            if (bv - bn1 < -E && gap > 2.0 * E) { upper = up ? 1 : 0; return bi; }                  // violated for certain, its side certain
        }
    }
    if (kProfile && w.prof && lane == 0) w.prof[15] += 1;      // (probe: scans the image left undecided)
    return rscan_rows_stream(w, upper);
}

template <int NB, int NP, bool FM, int IMG>
__device__ __forceinline__ void rrefine_active(RWave<NB, NP, FM, IMG> &w) // auxiliary.c:498-593
{
    const int lane = lane_id(), na = w.na, n = w.n;
    w.reuse = 0;
    double rhs = 0;
    if (lane < na) {
        const double *row = w.rowc + (size_t)w.slot * w.ldr;
        double mu = 0;
        for (int j = (w.wsid < w.ms ? w.wsid : 0); j < n; ++j) mu += row[j] * w.u[j];
        rhs = mu - (-w.drhs);                     // d = -drhs exactly
        if (w.wflag & DAQP_SOFT) rhs -= w.rho_soft * w.lams;
    }
    w.xl = rforward(w, w.xl, rhs, 0);
    if (lane < na) w.zl = w.xl / w.D;
    const double dlt = rbackward(w, (lane < na) ? w.zl : 0.0, na);
    if (lane < na) { w.xl = dlt; w.lams += dlt; }
    double uu = (lane < n) ? w.u[lane] : 0.0;
    for (int i = 0; i < na; ++i) {
        const double di = rl(w.xl, i);
        const int id = rli(w.wsid, i), s = rli(w.slot, i);
        const int j0 = id < w.ms ? id : 0;
        if (lane < n && lane >= j0) uu -= w.rowc[(size_t)s * w.ldr + lane] * di;
    }
    WSYNC();
    if (lane < n) w.u[lane_now()] = uu;
    WSYNC();
    double fv = w.soft;
    for (int j = 0; j < n; ++j) { const double uj = w.u[j]; fv += uj * uj; }
    w.fval = fv;
}

template <int NB, int NP, bool FM, int IMG>
__device__ __forceinline__ void rreset_ws(RWave<NB, NP, FM, IMG> &w) { w.sing = kEmpty; w.na = 0; w.reuse = 0; w.slotmask = 0; }

template <int NB, int NP, bool FM, int IMG>
__device__ __forceinline__ unsigned long long active_mask(const RWave<NB, NP, FM, IMG> &w, int bb)
{
    return __ballot(bb * 64 + lane_id() < w.m && (rsense_get(w, bb) & DAQP_ACTIVE));
}

// ---------------------------------------------------------------------------------------
// daqp_ldp (daqp.c:6-108) + daqp_activate_constraints (auxiliary.c:399-479) + daqp_pivot_last
// (auxiliary.c:379-396) as ONE explicit state machine.  The reference reaches add_constraint /
// remove_constraint from a dozen places and recurses through pivot_last; inlining that call
// tree multiplied the code of every primitive (and its register pressure) tenfold, so here each
// primitive has exactly one call site and "who asked" is a continuation code.
//   mode 1: only rebuild the working set from the ACTIVE bits (tail of daqp_update_ldp).
// Returns the exit flag (mode 0) or the activation flag (mode 1).
// ---------------------------------------------------------------------------------------
enum : int { PC_START_LOOP, PC_ITER, PC_EDIT, PC_ACT_BEGIN, PC_ACT_NEXT, PC_ACT_POST, PC_DONE };
enum : int { AFTER_NEXT_ITER, AFTER_CYCLE_GUARD, AFTER_ACT_POST };   // what follows a completed working-set edit
enum : int { ACT_THEN_DONE, ACT_THEN_LOOP, ACT_THEN_NEXT_ITER, ACT_THEN_CYCLE_RESET };

template <int NB, int NP, bool FM, int IMG>
__device__ __forceinline__ int rrun(RWave<NB, NP, FM, IMG> &w, int mode, bool need_activate, int &iterations)
{
    const int lane = lane_id();
    int flag = DAQP_EXIT_ITERLIMIT, it = 1, repaired = 0, stall = 0;
    double best = -1;
    const double fbound = 2 * w.stp->fval_bound;
    const int iter_limit = __builtin_amdgcn_readfirstlane(w.stp->iter_limit);
    // read once: inside the loop these would be two dependent trips to memory per iteration (the stores in the loop keep
    // the compiler from hoisting them itself)
    const double progress_tol = rl(w.stp->progress_tol, 0);
    const int cycle_tol = __builtin_amdgcn_readfirstlane(w.stp->cycle_tol);
    // settings->time_limit > 0 (daqp.c:95-103): every 32nd iteration that reaches the end of the reference's loop body looks
    // at the clock; the start stamp sits in LDS (u[66], see k_ldp_reg) so that it costs no register across the loop
    const bool tl_armed = rl(w.stp->time_limit, 0) > 0.0;
    int tl_skip = 0;   // this iteration ends with one of the reference's `continue`s: no clock check
#define RTL_CHECK() (tl_armed && !tl_skip && (it & 31) == 0 && time_is_up(reinterpret_cast<const unsigned long long *>(w.u)[66], rl(w.stp->time_limit, 0), rl(w.u[67], 0)))
    // edit request (add / drop, then the pivot_last cascade) and its continuation
    int depth = 0, req_add = 1, req_id = 0, req_r = 0, after_edit = AFTER_NEXT_ITER;
    double req_lam = 0;
    // activation cursor
    int act_then = ACT_THEN_DONE, act_bb = 0, act_i = 0, act_flag = 1;
    unsigned long long act_msk = 0;
    int pc;
    if (mode == 1 || need_activate) { rreset_ws(w); act_then = (mode == 1) ? ACT_THEN_DONE : ACT_THEN_LOOP; pc = PC_ACT_BEGIN; }
    else pc = PC_START_LOOP;
    while (pc != PC_DONE) {
        const int pc_now = pc;
        const long long t_in = (kProfile && w.prof) ? (long long)__builtin_readcyclecounter() : 0;
        switch (pc) {
        case PC_START_LOOP:
            if (act_flag < 0) { flag = act_flag; pc = PC_DONE; break; }
            it = 1;
            pc = (it < iter_limit) ? PC_ITER : PC_DONE;
            break;
        // ---- one iteration of daqp_ldp up to its working-set edit (daqp.c:12-64, 86-93)
        case PC_ITER: {
            tl_skip = 0;
            const bool was_singular = (w.sing != kEmpty);
            RPROF_T0(w);
            if (!was_singular) rsolve_csp(w); else { rtrace(w, kTraceSingular); rsingular_direction(w); }
            RPROF_ACC(w, 7);
            const int blk = rblocking_test(w);
            RPROF_ACC(w, 8);
            if (blk != kBig) { req_add = 0; req_r = blk; depth = 0; after_edit = AFTER_NEXT_ITER; pc = PC_EDIT; break; }
            if (was_singular) { flag = DAQP_EXIT_INFEASIBLE; pc = PC_DONE; break; }
            rprimal_u(w);
            RPROF_ACC(w, 9);
            int upper = 0;
            int pick;
            if constexpr (IMG != 0) pick = rscan_rows_img(w, upper); else pick = rscan_rows(w, upper, true);
            RPROF_ACC(w, 10);
            if (w.fval > fbound) { flag = DAQP_EXIT_INFEASIBLE; pc = PC_DONE; break; }
            after_edit = AFTER_CYCLE_GUARD;
            if (pick == kBig) {
                const double dmin = (w.na > 0) ? wave_min(lane < w.na ? w.D : (double)DAQP_INF) : (double)DAQP_INF;
                if (w.na > 2 && repaired != 1 && dmin < w.stp->refactor_tol) {
                    repaired = 1;
                    tl_skip = 1;
                    rtrace(w, kTraceRefactor);
                    for (int i = 0; i < w.na; ++i) {
                        const int id = rli(w.wsid, i);
                        if (rl(w.lam, i) >= 0) sense_set(w, id, 0, DAQP_LOWER); else sense_set(w, id, DAQP_LOWER, 0);
                    }
                    rreset_ws(w);
                    act_then = ACT_THEN_NEXT_ITER; pc = PC_ACT_BEGIN;
                    break;
                }
                if (w.na > 0 && dmin < w.pivot_tol) {
                    if constexpr (IMG != 0) {    // (the refinement step reads its rows lane <-> row from LDS: with rows in the scratch tier the problem goes to the kernel behind)
                        if (__ballot(lane < w.na && w.slot >= w.cache_slots)) { flag = kRegHandOverFlag; pc = PC_DONE; break; }
                    }
                    rtrace(w, kTraceRefine);
                    rrefine_active(w);
                    if constexpr (IMG != 0) {      // (rare: the refined u decides in fp64; u32 is not refreshed)
                        pick = rscan_rows_stream(w, upper);
                    } else pick = rscan_rows(w, upper, false);
                    after_edit = AFTER_NEXT_ITER;
                    tl_skip = 1;
                }
                if (pick == kBig) {
                    flag = (w.soft > w.stp->primal_tol) ? DAQP_EXIT_SOFT_OPTIMAL : DAQP_EXIT_OPTIMAL;
                    pc = PC_DONE;
                    break;
                }
            }
            // auxiliary.c:152-166: fix the side, lam <-> lam*, then add with multiplier +-1
            if (upper) sense_set(w, pick, 0, DAQP_LOWER); else sense_set(w, pick, DAQP_LOWER, 0);
            { const double t = w.lam; w.lam = w.lams; w.lams = t; }
            req_add = 1; req_id = pick; req_lam = upper ? 1.0 : -1.0; depth = 0; pc = PC_EDIT;
            break;
        }
        // ---- add_constraint / remove_constraint with the pivot_last cascade (auxiliary.c:3-44, 379-396),
        // each primitive instantiated once; then the caller's continuation
        case PC_EDIT: {
            // the shape that serves n = 64 (cap = n + 1 = 65 rows, one more than there are lanes): an add beyond the rows this kernel can
            // hold leaves the problem as it was stored and flags it for the one-wave generic kernel (reg_kernel.hip.h, launch_ldp).  The
            // re-adds of the pivot cascade below follow a removal and never exceed the level of the add that started it.
            if (kRegHandOver<NB, NP, IMG> && req_add && w.na >= w.max_rows) { flag = kRegHandOverFlag; pc = PC_DONE; break; }
            for (;;) {
                bool settled = false;
                RPROF_T0(w);
                if (req_add) { rpush_core(w, req_id, req_lam); RPROF_ACC(w, 13); }
                else { settled = rdrop_core(w, req_r) != 0; RPROF_ACC(w, 14); }     // a removal that left a singular factor does not pivot
                if (!settled) {
                    const int r = w.na - 2;
                    bool piv = false;
                    if (w.na > 1) {
                        const double dr = rl(w.D, r), dlast = rl(w.D, w.na - 1);
                        piv = dr < w.pivot_tol && dr < dlast;
                    }
                    if (piv) {
                        rtrace(w, kTracePivot);
                        const int idp = rli(w.wsid, r);
                        const double lp = rl(w.lam, r);
                        if (lane == 0) { w.pend_id[depth] = idp; w.pend_lam[depth] = lp; }
                        depth++;
                        WSYNC();
                        req_add = 0; req_r = r;
                        continue;
                    }
                    if (depth > 0 && w.sing == kEmpty) {
                        depth--;
                        req_id = __builtin_amdgcn_readfirstlane(w.pend_id[depth]); req_lam = rl(w.pend_lam[depth], 0);
                        req_add = 1;
                        continue;
                    }
                }
                break;
            }
            if (after_edit == AFTER_ACT_POST) { pc = PC_ACT_POST; break; }
            if (after_edit == AFTER_CYCLE_GUARD) {   // daqp.c:66-85
                if (w.fval - best < progress_tol) {
                    if (stall++ > cycle_tol) {
                        if (repaired == 1) { flag = DAQP_EXIT_CYCLE; pc = PC_DONE; break; }
                        repaired = 1;
                        rtrace(w, kTraceCycleReset);
                        rreset_ws(w);
                        act_then = ACT_THEN_CYCLE_RESET; pc = PC_ACT_BEGIN;
                        break;
                    }
                } else { best = w.fval; stall = 0; }
            }
            if (RTL_CHECK()) { flag = DAQP_EXIT_TIMELIMIT; pc = PC_DONE; break; }
            ++it;
            pc = (it < iter_limit) ? PC_ITER : PC_DONE;   // falling out of the loop: flag stays ITERLIMIT
            break;
        }
        // ---- daqp_activate_constraints: ACTIVE rows in index order (auxiliary.c:399-479)
        case PC_ACT_BEGIN:
            act_bb = 0; act_flag = 1;
            act_msk = active_mask(w, 0);
            pc = PC_ACT_NEXT;
            break;
        case PC_ACT_NEXT: {
            while (act_msk == 0 && act_bb + 1 < NB) { act_bb++; act_msk = active_mask(w, act_bb); }
            if (act_msk == 0) {   // done: continue where the activation was requested from
                if (act_then == ACT_THEN_DONE) pc = PC_DONE;
                else if (act_then == ACT_THEN_LOOP) pc = PC_START_LOOP;
                else {
                    if (act_then == ACT_THEN_CYCLE_RESET) { stall = 0; best = -1; }
                    if (RTL_CHECK()) { flag = DAQP_EXIT_TIMELIMIT; pc = PC_DONE; break; }
                    ++it;
                    pc = (it < iter_limit) ? PC_ITER : PC_DONE;
                }
                break;
            }
            act_i = act_bb * 64 + __ffsll((long long)act_msk) - 1;
            act_msk &= act_msk - 1;
            req_add = 1; req_id = act_i; req_lam = (sense_of(w, act_i) & DAQP_LOWER) ? -1.0 : 1.0;
            depth = 0; after_edit = AFTER_ACT_POST; pc = PC_EDIT;
            break;
        }
        case PC_ACT_POST: {
            if (w.sing == kEmpty) { pc = PC_ACT_NEXT; break; }
            const int lastflag = rli(w.wflag, w.na - 1);
            const int last = rli(w.wsid, w.na - 1);
            bool stop = true;
            if (lastflag & DAQP_IMMUTABLE) {   // a new equality depends on the active ones
                rsingular_direction(w);
                double resid = 0.0, scale = 1.0;
                const double term = (lane < w.na) ? w.lams * (-w.drhs) : 0.0;
                for (int j = 0; j < w.na; ++j) {
                    const double t = rl(term, j);
                    resid += t;
                    scale += t < 0 ? -t : t;
                }
                sense_set(w, last, 0, DAQP_ACTIVE);
                w.slotmask &= ~(1ull << rli(w.slot, w.na - 1));
                w.na--;
                w.sing = kEmpty;
                if (w.reuse > w.na) w.reuse = w.na;
                if (resid <= w.stp->primal_tol * scale && resid >= -w.stp->primal_tol * scale) { pc = PC_ACT_NEXT; break; }
                act_flag = DAQP_EXIT_OVERDETERMINED_INITIAL;
            } else {
                int fl = 1;
                static_for<NB>([&](auto b2) __attribute__((always_inline)) {   // rows >= i: unactivated equalities are an error, the rest are cleaned
                    const int r = b2 * 64 + lane;
                    const int sn = rsense_get(w, b2);
                    const bool later = r >= act_i && r < w.m && (sn & DAQP_ACTIVE);
                    if (__any(later && (sn & DAQP_IMMUTABLE))) fl = DAQP_EXIT_OVERDETERMINED_INITIAL;
                    if (later && !(sn & DAQP_IMMUTABLE)) w.rs &= ~((typename RWave<NB, NP, FM, IMG>::rs_t)DAQP_ACTIVE << (8 * b2));
                });
                w.slotmask &= ~(1ull << rli(w.slot, w.na - 1));
                w.na--;
                w.sing = kEmpty;
                act_flag = fl;
            }
            if (stop) {   // activation ends here (with act_flag); same continuations as a completed scan of the rows
                act_msk = 0; act_bb = NB;
                pc = PC_ACT_NEXT;
            }
            break;
        }
        default:
            pc = PC_DONE;
            break;
        }
        if (kProfile && w.prof) {   // cycles and visits per state
            const long long t_out = (long long)__builtin_readcyclecounter();
            if (lane == 0) { w.prof[pc_now] += t_out - t_in; w.prof[16 + pc_now] += 1; }
        }
    }
#undef RTL_CHECK
    iterations = it;
    return (mode == 1) ? act_flag : flag;
}

} // namespace daqp_amd
