// reg_kernel.hip.h -- k_ldp_reg: the register-centric solve kernel (wave_ldp_reg.hip.h) for n + n_soft + 1 <= 64 and
// m <= 64*NB, n <= 2*NP.  Its own translation unit (reg_kernel.hip) instantiates the shapes; the host code only needs the
// LDS sizes and the declarations.
#pragma once
#include "wave_ldp_reg.hip.h"
#include "batch_dev.hip.h"

namespace daqp_amd {

// LDS of the register-centric solve kernel.  Everything small sits at COMPILE-TIME offsets in front (u, the pivot
// stack, the per-row bounds, packed L): their addresses are "immediate + 8*lane", nothing to keep in a register across
// the state-machine loop (every loop-invariant base pointer is one more value that the full register file spills to
// scratch).  Only the active-row cache, behind the run-time sized L, has a run-time base.
template <int NB, int IMG = 0>
struct RegLds {
    static constexpr int u = 0, pend_lam = 68, pend_id = 132, prof = 164, u32 = 196, rowv = 196 + (IMG ? 32 : 0), L = rowv + 3 * kRowvStride<NB, IMG>;   // doubles (u32: 64 floats, IMG != 0 only)
};
__host__ __device__ inline int reg_lds_rowc_size(int n, int m, int cap, int ldrc)
{
    const int rows = round_up(cap * ldrc, 2), fin = round_up(n * (n + 1) / 2, 2) + 2 + round_up(m, 2);   // epilogue: staged R^-1 + lam
    return rows > fin ? rows : fin;
}
// (IMG = 1: `cap` is the number of working-set rows the kernel holds -- BatchDev::img_rows --, not the problem's own cap)
__host__ __device__ inline int reg_lds_rowc(int NB, int cap, int IMG = 0) { return 196 + (IMG ? 32 : 0) + 3 * (IMG == 2 ? 64 * NB - 32 : 64 * NB) + round_up(cap * (cap + 1) / 2, 2); }
__host__ __device__ inline int reg_lds_bytes(int NB, int n, int m, int cap, int ldrc, int IMG = 0) { return 8 * (reg_lds_rowc(NB, cap, IMG) + reg_lds_rowc_size(n, m, cap, ldrc)); }
// IMG != 0: [front: u, pivot stack, probes, u32][row view][L: tri(rows)][row cache], the cache tiered (RWave::cache_slots): `cache` rows + one
// staging row when cache < rows, else all `rows`; L and the cache together at least as large as the epilogue's staging area (R^-1 + lam), which
// starts at L (the factor has been stored by then)
__host__ __device__ inline int reg_img_cache_rows(int rows, int cache) { return cache < rows ? cache + 1 : rows; }
__host__ __device__ inline int reg_img_stage_size(int n, int m, int rows, int cache, int ldrc)
{
    const int body = round_up(rows * (rows + 1) / 2, 2) + round_up(reg_img_cache_rows(rows, cache) * ldrc, 2), fin = round_up(n * (n + 1) / 2, 2) + 2 + round_up(m, 2);
    return body > fin ? body : fin;
}
__host__ __device__ inline int reg_img_lds_bytes(int NB, int IMG, int n, int m, int rows, int cache, int ldrc)
{
    return 8 * (196 + 32 + 3 * (IMG == 2 ? 64 * NB - 32 : 64 * NB) + reg_img_stage_size(n, m, rows, cache, ldrc));
}

// ------------------------------------------------------------------------------------
// k_ldp_reg: the register-centric solve kernel (wave_ldp_reg.hip.h) for n + n_soft + 1 <= 64 and
// m <= 64*NB, n <= 2*NP.  Same global state layout as k_ldp, so the two are interchangeable.
// ------------------------------------------------------------------------------------
// Waves per SIMD: M alone takes 4*NB*NP registers per lane.  The large shapes own the whole unified 512-entry file (one
// wave per SIMD); the small ones are held to a budget that lets 2 or 4 waves share a SIMD, where the other waves'
// instructions fill this wave's issue gaps and LDS waits.
// (small shapes: 2 -> 3 waves per SIMD is 1.36x on config C3, 3 -> 4 another 1.05x although 17-51 registers then live in scratch
// (128-register budget); n = 16, m = 64: 1.05x, n = 14, m = 40: unchanged -- profiles/r05w_small_shape_waves.txt)
#ifndef DAQP_AMD_SMALL_WAVES
#define DAQP_AMD_SMALL_WAVES 4
#endif
// (NB*NP <= 13, i.e. n <= 26 and m <= 64: three waves per SIMD -- twelve waves' LDS still fit a CU up to there, and the same code on
// the (1,16) shape measured 16 % faster at three waves than at two for n = 20 / 24, 2 % slower for n = 30 / 32 where the LDS does not fit)
// (IMG = 1: the large shapes with an fp32 image of M instead of M -- 2 NB NP registers -- at two waves per SIMD)
// (IMG = 1 with more than 96 image pairs -- (4,32): n <= 63, m <= 256 --: one wave per SIMD again, but shapes that have NO full-register kernel at all:
//  their M itself would be 512 registers.  Before round 6's last session they ran the one-wave kernel that streams M from HBM / L2 in every scan)
constexpr int ldp_reg_waves(int NB, int NP, int IMG = 0) { return IMG ? (NB * NP > 96 ? 1 : 2) : (NB * NP <= 8 ? DAQP_AMD_SMALL_WAVES : (NB * NP <= 13 ? 3 : (NB * NP <= 32 ? 2 : 1))); }
template <int NB, int NP, bool FM, int IMG = 0>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(ldp_reg_waves(NB, NP, IMG), ldp_reg_waves(NB, NP, IMG)))) void k_ldp_reg(const BatchDev *__restrict__ bp, int mode_in)
{
    static_assert(IMG == 0 || FM, "the fp32 image screens the scan of the default arithmetic only");
    // mode 0: daqp_solve; 1: only (re)build the working set from the ACTIVE bits; 2 | mask << 4: daqp_update_ldp(mask)
    // for mask within UPDATE_v|UPDATE_d applied here, then daqp_solve -- the rows of M are in registers anyway, so the
    // warm path of an MPC step reads them from HBM once instead of twice (k_update + solve)
    // | 4: only the problems an IMG = 1 launch in front flagged in `fallback` (more working-set rows than its LDS holds)
    const int upd = (mode_in & 3) == 2 ? ((mode_in >> 4) & 0xfff) : 0;
    const int mode = (mode_in & 3) == 2 ? 0 : (mode_in & 3);
    // The descriptor is read through a pointer (scalar loads at the point of use) instead of being a
    // by-value kernel argument: ~60 SGPRs of pointers would otherwise stay live across the whole state
    // machine and push its uniform state into VGPR-lane spills.
    const BatchDev &b = *bp;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const long long t_start = (long long)__builtin_readcyclecounter();
    const int q = blockIdx.x, lane = lane_id();
    const int n = b.n, m = b.m, cap = b.cap;
    if (mode_in & 4) { if (b.fallback == nullptr || !__builtin_amdgcn_readfirstlane(as_global(b.fallback)[q])) return; }
    // (a first pass finds every flag cleared: launch_ldp zeroes the array in front of it.  NOT here: a store pending at this point makes every
    //  wait for the row loads below a wait for everything -- loads and stores share the counter and may retire out of order --, and the
    //  double-buffered stream of the image degenerates into one memory round trip per unit)
    // rows of the working set the LDS carve-up is sized for: the problem's own cap, or (IMG = 1) what the host chose to keep two waves per SIMD
    // (IMG != 0: a launch may bring its own carve-up -- rows << 16 | cache << 22 in the launch argument, see launch_ldp: warm launches hold fewer rows)
    const int lds_rows = IMG ? (((mode_in >> 16) & 63) ? ((mode_in >> 16) & 63) : __builtin_amdgcn_readfirstlane(b.img_rows)) : cap;
    DAQP_GLOBAL(QState) *qs = as_global(b.qs + q);   // (global pointers throughout: see DAQP_GLOBAL in wave_ldp.hip.h)
    if (mode == 1) {   // an activation launch looks at the record first: almost every problem leaves here, without touching M
        if (__builtin_amdgcn_readfirstlane(qs->setup_flag) < 0 || !__builtin_amdgcn_readfirstlane(qs->need_activate)) return;
    }
    typedef RegLds<NB, IMG> o;
    // ---- ONE batch of loads.  Everything whose address does not depend on the per-problem record goes out before the first
    // wait: the rows of M, the row view, the new bounds / f / R^-1 of a pending update, the working-set ids and vectors of a
    // warm start -- and the record itself (before: the record, then the ids, then M -- three dependent trips).  Measured on the
    // warm path (tools/c5_phases.py): issuing this batch takes 11.6 k cycles -- the 76.8 KB of M at what one CU's miss path
    // delivers to four waves -- and the record is there when the last load has been issued; prologue 35.2 k -> 34.1 k of a warm
    // solve's 141 k cycles (then: active rows + L 3.5 k, v = R^-T f 5.6 k, d 2.5 k).
    const size_t qfac = qf(b, q);
    DAQP_GLOBAL(double) *gdu = as_global(b.dupper + (size_t)q * m), *gdl = as_global(b.dlower + (size_t)q * m);
    const DAQP_GLOBAL(double) *gsc = as_global(b.scaling + qfac * m);
    DAQP_GLOBAL(int) *gsense = as_global(b.sense + (size_t)q * m);
    DAQP_GLOBAL(double) *gv = as_global(b.vecs + (size_t)q * 5 * cap);
    DAQP_GLOBAL(int) *gws = as_global(b.WS + (size_t)q * cap);
    const int img_cache = IMG ? (((mode_in >> 16) & 63) ? ((mode_in >> 22) & 63) : __builtin_amdgcn_readfirstlane(b.img_cache)) : 0;
    const int rowc_size = IMG ? reg_img_stage_size(n, m, lds_rows, img_cache, b.ldrc) : reg_lds_rowc_size(n, m, lds_rows, b.ldrc);
    RWave<NB, NP, FM, IMG> w;
    w.rowc = smem + (IMG ? o::L + round_up(lds_rows * (lds_rows + 1) / 2, 2) : reg_lds_rowc(NB, lds_rows, IMG));
    // staging area of R^-1 (pending UPDATE_v; epilogue) and of lam: the row cache -- IMG != 0: L and the row cache (the update's v is formed before
    // the factor is loaded, the epilogue stores the factor first)
    double *sbase = IMG ? smem + o::L : w.rowc;
    // a pending UPDATE_v needs R^-1 and f: their loads go out first and arrive together with the rows of M
    const int rinv0 = rowc_size - round_up(b.rtri, 2) - 2;
    double *Rl0 = sbase + rinv0;
    double f_raw = 0, f_sc = 1;
    if (upd & DAQP_UPDATE_v) {
        const double *Rq = b.Rinv + qfac * b.rtri, *f = b.f + (size_t)q * n;
        const int odd8 = (int)(((size_t)Rq >> 3) & 1);
        Rl0 += odd8;
        if (odd8) copy_async_dwords(Rl0, Rq, 1);
        const int body = (b.rtri - odd8) & ~1;
        copy_async(Rl0 + odd8, Rq + odd8, body);
        if (odd8 + body < b.rtri) copy_async_dwords(Rl0 + odd8 + body, Rq + odd8 + body, 1);
        if (lane < n) { f_raw = as_global(f)[lane]; if (lane < b.ms) f_sc = gsc[lane]; }
    }
    // ---- row view: bounds, tolerance, sense and the rows of M themselves -> registers
    double scr[NB];
    int softbits = 0;
    w.rs = 0;
    typedef double gv2d __attribute__((ext_vector_type(2)));
    const DAQP_GLOBAL(gv2d) *msrc = as_global(reinterpret_cast<const gv2d *>(b.Mblk + qfac * b.nblk * b.npair * 128));
    const int npair_u = __builtin_amdgcn_readfirstlane(b.npair), nblk_u = __builtin_amdgcn_readfirstlane(b.nblk);
    // the small per-row loads go out first: in-order return means whoever waits for them would otherwise wait for
    // every row of M issued before them
    double dur[NB], dlr[NB];
    int snr[NB];
    static_for<NB>([&](auto bb) __attribute__((always_inline)) {
        const int r = bb * 64 + lane;
        const bool ok = r < m;
        scr[bb] = ok ? gsc[r] : 0.0;
        dur[bb] = ok ? gdu[r] : 0.0;
        dlr[bb] = ok ? gdl[r] : 0.0;
        snr[bb] = ok ? (gsense[r] & 0xff) : 0;
    });
    // a pending update also needs the new bounds: same early batch
    double bur[NB], blr[NB];
    if (upd) {
        const DAQP_GLOBAL(double) *nbu = as_global(b.bu + (size_t)q * m), *nbl = as_global(b.bl + (size_t)q * m);
        static_for<NB>([&](auto bb) __attribute__((always_inline)) {
            const int r = bb * 64 + lane;
            bur[bb] = (r < m) ? nbu[r] : 0.0;
            blr[bb] = (r < m) ? nbl[r] : 0.0;
        });
    }
    // the warm-start state: working-set ids / vectors into registers (masked by n_active once the record is here)
    const int wsid_raw = (lane < cap) ? gws[lane] : 0;
    const double D_r = (lane < cap) ? gv[lane] : 0.0, xl_r = (lane < cap) ? gv[cap + lane] : 0.0, zl_r = (lane < cap) ? gv[2 * cap + lane] : 0.0;
    const double la_r = (lane < cap) ? gv[3 * cap + lane] : 0.0, lb_r = (lane < cap) ? gv[4 * cap + lane] : 0.0;
    // the record (wave-uniform: readfirstlane'd below, after everything else has been issued)
    const int rec_sflag = qs->setup_flag, rec_need = qs->need_activate, rec_diag = qs->diag_h, rec_uflag = qs->upd_flag;
    const int rec_na = qs->n_active, rec_reuse = qs->reuse_ind, rec_sing = qs->sing_ind, rec_swapped = qs->lam_swapped;
    const double rec_fval = qs->fval, rec_soft = qs->soft_slack;
    // v = R^-T f (utils.c:474-497) of a pending UPDATE_v, rows < ms of R^-1 being the normalised ones: from R^-1 and f staged in LDS
    auto v_of_update = [&](const double *Rl, const double *fl) __attribute__((always_inline)) -> double {
        if constexpr (FM) {
            // default arithmetic: the triangle FOLDED over the lanes.  Column i has i + 1 terms -- lane n-1 would run an n-step
            // chain while lane 0 runs one (5.6 k cycles of a warm solve at n = 50) -- so lane i takes the first LH = n/2 + 1 terms of
            // its own column (j = i, i-1, ...) and then the tail of column n-1-i that its owner leaves over (j = n-2-i-LH .. 0 when
            // that column is longer than LH): every lane ~n/2 terms, one exchange at the end.  Same products, one more association
            // than the reference's single chain (the exact mode keeps utils.c:474-497's order below).
            const int i = lane < n ? lane : 0, pc = n - 1 - i;                 // own column, partner column
            const int LH = n / 2 + 1;
            const int own_lo = (i + 1 > LH) ? i - LH + 1 : 0;                      // own terms: j = i .. own_lo
            const int tail_hi = (pc + 1 > LH && pc != i) ? pc - LH : -1;           // partner's left-over: j = tail_hi .. 0
            double acc = 0, acc2 = 0;
            for (int j0 = i; j0 >= own_lo; j0 -= kChunk) {
                double rr[kChunk], ff[kChunk];
#pragma unroll
                for (int k = 0; k < kChunk; ++k) { const int j = (j0 - k >= own_lo) ? j0 - k : own_lo; rr[k] = Rl[roff(j, n) + i]; ff[k] = fl[j]; }
#pragma unroll
                for (int k = 0; k < kChunk; ++k) if (j0 - k >= own_lo) acc = __builtin_fma(rr[k], ff[k], acc);
            }
            for (int j0 = tail_hi; j0 >= 0; j0 -= kChunk) {
                double rr[kChunk], ff[kChunk];
#pragma unroll
                for (int k = 0; k < kChunk; ++k) { const int j = (j0 - k >= 0) ? j0 - k : 0; rr[k] = Rl[roff(j, n) + pc]; ff[k] = fl[j]; }
#pragma unroll
                for (int k = 0; k < kChunk; ++k) if (j0 - k >= 0) acc2 = __builtin_fma(rr[k], ff[k], acc2);
            }
            const double other = __shfl(acc2, (lane < n) ? n - 1 - lane : lane);   // the lane that worked on THIS lane's column
            return acc + other;
        } else {
            const int i = lane < n ? lane : 0;
            double acc = Rl[roff(i, n) + i] * fl[i];
            for (int j0 = i - 1; j0 >= 0; j0 -= kChunk) {
                double rr[kChunk], ff[kChunk];
#pragma unroll
                for (int k = 0; k < kChunk; ++k) { const int j = (j0 - k >= 0) ? j0 - k : 0; rr[k] = Rl[roff(j, n) + i]; ff[k] = fl[j]; }
#pragma unroll
                for (int k = 0; k < kChunk; ++k) if (j0 - k >= 0) acc += rr[k] * ff[k];
            }
            return acc;
        }
    };
    double v_early = 0;
    bool v_done = false;
    __builtin_amdgcn_sched_barrier(0);
    double sm_img[NB];      // IMG = 1: row . v of a pending update, formed while the rows pass through
    static_for<NB>([&](auto bb) __attribute__((always_inline)) { sm_img[bb] = 0; });
    if constexpr (IMG != 0) {
        // The fp64 rows pass through the registers once: each pair is rounded into the image as it arrives, and a pending update's
        // d = b s + M v (utils.c:499-544) takes its row . v from the passing fp64 values -- so v must be in LDS BEFORE the stream.
        w.msrc = msrc; w.npair = npair_u; w.u32 = reinterpret_cast<float *>(smem + o::u32);
        w.cache_slots = img_cache;
        w.rowg = as_global(b.rowc_g + (size_t)q * (size_t)((lds_rows > img_cache ? lds_rows - img_cache : 0) * b.ldrc));
        if (upd) {
            // A warm launch looks at the stored working set FIRST: one that is within two rows of what this launch holds will outgrow it with the
            // next constraints (an MPC step moves a few rows), so it goes to the full-register kernel now -- before 60 KB of M have been streamed
            // for nothing.  (What a long drift of f does to C5's working sets -- 43 of 50 rows active after a hundred steps -- is in tools/c5_walk.py:
            // such batches run at the full-register kernel's rate, not below it.)
            copy_wait();
            if (b.fallback != nullptr && __builtin_amdgcn_readfirstlane(rec_na) + 2 > lds_rows) {
                if (lane == 0) { as_global(b.fallback)[q] = 1; if (b.img_ho) atomicAdd(b.img_ho, 1); }
                // (wait for them HERE: a store and a flat atomic left pending on this -- returning -- path still reach the code generator's
                //  picture of the stream below, where loads mixed with pending stores can only be waited for all at once: vmcnt(0) at every
                //  unit instead of vmcnt(19), (18), ... ; with the wait the two units in flight really overlap: C5 22.8 -> 23.3 M warm solves/s)
                __builtin_amdgcn_s_waitcnt(0);
                return;
            }
            double *vv = smem + o::u, *fl = smem + o::pend_lam;
            for (int e = lane; e < round_up(n > 64 ? n : 64, 2) + 2; e += 64) vv[e] = 0;
            if (upd & DAQP_UPDATE_v) {
                copy_wait();
                const int qdiag_e = __builtin_amdgcn_readfirstlane(rec_diag);
                if (lane < n) fl[lane] = (lane < b.ms && !qdiag_e) ? f_raw / f_sc : f_raw;
                WSYNC();
                v_early = v_of_update(Rl0, fl);
                v_done = true;
                if (lane < n) vv[lane] = v_early;
            } else if (lane < n) vv[lane] = as_global(b.v)[(size_t)q * n + lane];
            WSYNC();
        }
        const double2 *v2 = reinterpret_cast<const double2 *>(smem + o::u);
        // (GB = 10: the code generator waits for EVERYTHING outstanding at a unit's first use -- vmcnt(0), whatever is done about the stores and LDS copies
        //  around it --, so two units make one memory round trip: 20 pairs per trip instead of 12 took 4 % off the C2 launch and off a warm solve;
        //  13 pairs per unit start to spill)
        // Units of GB pairs, double-buffered: unit U + 1 is in flight while unit U is rounded into the image -- and no further: left to itself
        // the scheduler hoists ALL the loads (300 registers of fp64 temporaries), and image values defined under that pressure are spilled
        // for good (reloaded by every scan).  Full blocks first (lane <-> row), then (IMG = 2) the last block with two lanes per row: lane
        // 2k + h holds pairs NPH h .. NPH h + NPH - 1 of row 64 (NB-1) + k (wave_ldp_reg.hip.h rscan_rows_img).
#ifndef DAQP_IMG_GB
#define DAQP_IMG_GB 10
#endif
        constexpr int GB = DAQP_IMG_GB, NBF = kImgFullBlocks<NB, IMG>, NBAT = (NP + GB - 1) / GB, NPH = (NP + 1) / 2, NHB = (IMG == 2) ? (NPH + GB - 1) / GB : 0;
        constexpr int NUNITS = NBF * NBAT + NHB;
        const int h = lane & 1, k2 = lane >> 1;
        const bool rowok2 = (NB - 1) < nblk_u && 64 * (NB - 1) + k2 < m;
        const int rsel2 = rowok2 ? k2 : 0;
        double half = 0;
        gv2d buf[2][GB];
        auto load_unit = [&](auto U) __attribute__((always_inline)) {
            constexpr int u = decltype(U)::value;
            if constexpr (u < NBF * NBAT) {
                constexpr int bb = u / NBAT, g = u % NBAT;
                const int lsel = (64 * bb + lane < m) ? lane : 0;      // lanes beyond m re-read lane 0's line (never looked at: every use of a row is behind r < m)
                static_for<GB>([&](auto k) __attribute__((always_inline)) {
                    constexpr int t = GB * g + k;
                    if constexpr (t < NP) {
                        const bool ok = bb < nblk_u && t < npair_u;
                        buf[u & 1][k] = msrc[(ok ? ((size_t)bb * npair_u + t) : (size_t)0) * 64 + lsel];
                    }
                });
            } else if constexpr (u < NUNITS) {
                constexpr int g = u - NBF * NBAT;
                static_for<GB>([&](auto k) __attribute__((always_inline)) {
                    constexpr int tt = GB * g + k;
                    if constexpr (tt < NPH) {
                        const int t = NPH * h + tt;
                        const bool ok = rowok2 && t < npair_u;
                        buf[u & 1][k] = msrc[(ok ? ((size_t)(NB - 1) * npair_u + t) : (size_t)0) * 64 + rsel2];
                    }
                });
            }
        };
        auto use_unit = [&](auto U) __attribute__((always_inline)) {
            constexpr int u = decltype(U)::value;
            if constexpr (u < NBF * NBAT) {
                constexpr int bb = u / NBAT, g = u % NBAT;
                static_for<GB>([&](auto k) __attribute__((always_inline)) {
                    constexpr int t = GB * g + k;
                    if constexpr (t < NP) {
                        const bool ok = bb < nblk_u && t < npair_u;
                        const double mx = ok ? buf[u & 1][k].x : 0.0, my = ok ? buf[u & 1][k].y : 0.0;
                        if (upd) { const double2 vk = v2[t]; sm_img[bb] = __builtin_fma(mx, vk.x, sm_img[bb]); sm_img[bb] = __builtin_fma(my, vk.y, sm_img[bb]); }
                        // (pinned: the rounding happens HERE -- left alone, it is sunk to the loop's entry and the fp64 pairs stay live across the prologue)
                        float fx = (float)mx, fy = (float)my;
                        asm volatile("" : "+v"(fx), "+v"(fy));
                        w.Mx[bb][t] = fx; w.My[bb][t] = fy;
                    }
                });
            } else {
                constexpr int g = u - NBF * NBAT;
                static_for<GB>([&](auto k) __attribute__((always_inline)) {
                    constexpr int tt = GB * g + k;
                    if constexpr (tt < NPH) {
                        const int t = NPH * h + tt;
                        const bool ok = rowok2 && t < npair_u;
                        const double mx = ok ? buf[u & 1][k].x : 0.0, my = ok ? buf[u & 1][k].y : 0.0;
                        if (upd) { const double2 vk = v2[t < NP ? t : 0]; half = __builtin_fma(mx, vk.x, half); half = __builtin_fma(my, vk.y, half); }
                        float fx = (float)mx, fy = (float)my;
                        asm volatile("" : "+v"(fx), "+v"(fy));
                        w.Mx[NB - 1][tt] = fx; w.My[NB - 1][tt] = fy;
                    }
                });
            }
        };
        load_unit(std::integral_constant<int, 0>{});
        static_for<NUNITS>([&](auto U) __attribute__((always_inline)) {
            load_unit(std::integral_constant<int, decltype(U)::value + 1>{});
            __builtin_amdgcn_sched_barrier(0);
            use_unit(U);
            __builtin_amdgcn_sched_barrier(0);
        });
        if constexpr (IMG == 2) { if (upd) { half += __shfl_xor(half, 1); sm_img[NB - 1] = __shfl(half, (2 * lane) & 63); } }
    } else
    if (nblk_u == NB && npair_u == NP) {
        // the shape fills the template exactly (the benchmark's case): NB*NP unconditional loads in ONE basic block, each
        // straight into its final (mostly accumulation) register -- all in flight, one wait at the first use
        // (the lanes of the last block whose rows lie beyond m -- 42 of 64 at m = 150 -- re-read lane 0's line instead of their own slice of
        // the image's padding: the same instructions, a fifth fewer lines from HBM; those registers are never looked at: every use of a row is behind r < m)
        const int lane_last = (64 * (NB - 1) + lane < m) ? lane : 0;
        static_for<NB - 1>([&](auto bb) __attribute__((always_inline)) {
            static_for<NP>([&](auto t) __attribute__((always_inline)) {
                const gv2d v = msrc[((size_t)bb * NP + t) * 64 + lane];
                w.Mx[bb][t] = v.x; w.My[bb][t] = v.y;
            });
        });
        if constexpr (NB > 1 && (NB - 1) * NP <= 60) {
            // A pending UPDATE_v is worked off HERE, between the row loads: issuing them is what the wave waits for (each 1 KB load takes
            // ~220 cycles to be accepted: 16.8 k cycles for the 75 of config 5), and arithmetic placed between them runs while the memory
            // path digests the ones already queued -- v = R^-T f was 3-5 k cycles spent AFTER the last row had arrived.  Its inputs (R^-1
            // by LDS copy, f, the record) went out before every row load, so "all but the last (NB-1) NP operations done" covers them.
            if (upd & DAQP_UPDATE_v) {
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_waitcnt(0x0F70 | (((NB - 1) * NP) & 15) | ((((NB - 1) * NP) >> 4) << 14));   // vmcnt((NB-1) NP)
                __builtin_amdgcn_sched_barrier(0);
                double *fl = smem + o::pend_lam;
                const int qdiag_e = __builtin_amdgcn_readfirstlane(rec_diag);
                if (lane < n) fl[lane] = (lane < b.ms && !qdiag_e) ? f_raw / f_sc : f_raw;
                WSYNC();
                v_early = v_of_update(Rl0, fl);
                v_done = true;
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        static_for<NP>([&](auto t) __attribute__((always_inline)) {
            const gv2d v = msrc[((size_t)(NB - 1) * NP + t) * 64 + lane_last];
            w.Mx[NB - 1][t] = v.x; w.My[NB - 1][t] = v.y;
        });
    } else {
        // smaller problems in the same register shape: lines beyond the problem's are zeros (a select per load keeps
        // only a few of them in flight; these shapes are far from bandwidth-critical)
        static_for<NB>([&](auto bb) __attribute__((always_inline)) {
            static_for<NP>([&](auto t) __attribute__((always_inline)) {
                const bool ok = bb < nblk_u && t < npair_u;
                const gv2d v = msrc[(ok ? ((size_t)bb * npair_u + t) : (size_t)0) * 64 + lane];
                w.Mx[bb][t] = ok ? v.x : 0.0; w.My[bb][t] = ok ? v.y : 0.0;
            });
        });
    }
    __builtin_amdgcn_sched_barrier(0);
    // ---- the record.  Everything read from it is wave-uniform; say so, or every loop bounded by n_active becomes a
    // divergent (exec-masked) loop with readfirstlane waterfalls around v_readlane
    const int sflag = __builtin_amdgcn_readfirstlane(rec_sflag);
    int q_need_act = __builtin_amdgcn_readfirstlane(rec_need);
    const int qdiag = __builtin_amdgcn_readfirstlane(rec_diag);
    if (sflag < 0) {
        if (lane == 0) { as_global(b.exitflag)[q] = sflag; as_global(b.iter)[q] = 0; if (b.fval) as_global(b.fval)[q] = 0; if (b.soft) as_global(b.soft)[q] = 0; }
        copy_wait();   // nothing may still be landing in LDS when the workgroup ends
        return;
    }
    if (mode == 0 && !upd) {   // the last update failed its bound check: report that, keep the state (see k_update)
        const int uflag = __builtin_amdgcn_readfirstlane(rec_uflag);
        if (uflag < 0) {
            if (lane == 0) { as_global(b.exitflag)[q] = uflag; as_global(b.iter)[q] = 0; if (b.fval) as_global(b.fval)[q] = 0; if (b.soft) as_global(b.soft)[q] = 0; }
            return;
        }
    }
    {   // time_limit stamp and seconds per tick (rrun reads them from u[66], u[67])
        const unsigned long long ts = solve_stamp(b.tstart, q);
        if (lane == 0) { reinterpret_cast<unsigned long long *>(smem + o::u)[66] = ts; smem[o::u + 67] = b.tick_s; }
    }
    // phase counters live in the (otherwise unused) D/xl slots of the LDS carve-up
    w.prof = (kProfile && (b.prof != nullptr) && mode == 0) ? reinterpret_cast<long long *>(smem + o::prof) : nullptr;
    if (kProfile && w.prof && lane < 32) w.prof[lane] = 0;
    w.n = n; w.m = m; w.ms = b.ms; w.ldr = b.ldrc;
    w.L = smem + o::L; w.u = smem + o::u; w.pend_lam = smem + o::pend_lam;
    w.rowv = smem + o::rowv;
    w.pend_id = reinterpret_cast<int *>(smem + o::pend_id);
    w.stp = as_global(b.st_dev);
    w.dual_tol = b.st.dual_tol; w.sing_tol = b.st.sing_tol; w.pivot_tol = b.st.pivot_tol; w.rho_soft = b.st.rho_soft;
    w.trace = b.trace ? as_global(b.trace + (size_t)q * b.trace_cap) : nullptr;
    w.trace_cap = b.trace_cap; w.trace_len = 0;
    w.na = __builtin_amdgcn_readfirstlane(rec_na);
    if constexpr (kRegHandOver<NB, NP, IMG>) {
        // (no `fallback` array: nobody stands behind this launch -- the kernel keeps every problem, up to its 64 lanes)
        w.max_rows = IMG ? lds_rows : ((b.fallback != nullptr) ? b.reg_rows : 64);
        if (w.na > w.max_rows) {   // stored by the kernel behind this one with more rows than this one holds: its problem again
            if (lane == 0 && b.fallback != nullptr) { as_global(b.fallback)[q] = 1; if (IMG != 0 && b.img_ho) atomicAdd(b.img_ho, 1); }
            copy_wait();
            return;
        }
    }
    w.reuse = __builtin_amdgcn_readfirstlane(rec_reuse);
    w.sing = __builtin_amdgcn_readfirstlane(rec_sing);
    w.fval = rl(rec_fval, 0); w.soft = rl(rec_soft, 0);
    const int swapped = __builtin_amdgcn_readfirstlane(rec_swapped);

    if (w.sing == DAQP_UNCONSTRAINED_OPTIMAL && mode == 0 && !upd) {   // api.c:40-45 (an update resets sing_ind first)
        const DAQP_GLOBAL(double) *xu = as_global(b.xunc + (size_t)q * n), *vq = as_global(b.v + (size_t)q * n);
        if (b.x) for (int i = lane; i < n; i += 64) as_global(b.x)[(size_t)q * n + i] = xu[i];
        if (b.lam) for (int i = lane; i < m; i += 64) as_global(b.lam)[(size_t)q * m + i] = 0;
        double fv = 0;
        for (int i = 0; i < n; ++i) { const double vi = vq[i]; fv -= vi * vi; }
        fv *= 0.5;
        if (lane == 0) {
            as_global(b.exitflag)[q] = DAQP_EXIT_OPTIMAL; as_global(b.iter)[q] = 1;
            if (b.fval) as_global(b.fval)[q] = fv;
            if (b.soft) as_global(b.soft)[q] = 0;
            qs->iterations = 1; qs->fval = 0; qs->soft_slack = 0; qs->exitflag = DAQP_EXIT_OPTIMAL;
        }
        return;
    }
    const double f_in = (lane < b.ms && !qdiag) ? f_raw / f_sc : f_raw;
    const double ep = -w.stp->primal_tol;
    // packed L and (the ids are here now) the active rows straight into LDS -- one more trip, overlapped with the update's arithmetic
    const bool act = lane < w.na;
    const int wsid_r = act ? wsid_raw : 0;
    copy_async(w.L, b.L + (size_t)q * b.ltri, round_up(tri(w.na), 2));
    auto fetch_active_rows = [&]() __attribute__((always_inline)) {
        for (int i = 0; i < w.na; ++i) {
            const int id = rli(wsid_r, i);
            if (IMG != 0 && i >= img_cache) {   // a warm start with more rows than the LDS tier holds: this one goes to the scratch tier (lane <-> component)
                const DAQP_GLOBAL(double) *srow = reinterpret_cast<const DAQP_GLOBAL(double) *>(msrc + ((size_t)(id >> 6) * b.npair) * 64 + (id & 63));
                if (lane < n) w.rowg[(i - img_cache) * b.ldrc + lane] = srow[(size_t)(lane >> 1) * 128 + (lane & 1)];
                continue;
            }
            const DAQP_GLOBAL(gv2d) *src = msrc + ((size_t)(id >> 6) * b.npair) * 64 + (id & 63) + (size_t)lane * 64;
            if (lane < b.npair)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                                 (__attribute__((address_space(3))) void *)(w.rowc + (size_t)i * w.ldr), 16, 0, 0);
        }
    };
    const bool rows_early = IMG != 0 || !(upd & DAQP_UPDATE_v) || w.na * w.ldr <= rinv0;   // R^-1 is staged in the top of the row cache (IMG != 0: not any more: v was formed first)
    if (rows_early) fetch_active_rows();
    static_for<NB>([&](auto bb) __attribute__((always_inline)) {
        const int r = bb * 64 + lane;
        if (IMG != 2 || r < kRowvStride<NB, IMG>) {     // (IMG = 2: the last block's part of each array is 32 rows long)
            w.rowv[r] = dur[bb];
            w.rowv[kRowvStride<NB, IMG> + r] = dlr[bb];
            w.rowv[2 * kRowvStride<NB, IMG> + r] = ep * scr[bb];
        }
        w.rs |= (typename RWave<NB, NP, FM, IMG>::rs_t)snr[bb] << (8 * bb);
        softbits |= snr[bb] & DAQP_SOFT;
    });
    w.has_soft = __any(softbits) ? 1 : 0;
    const long long tp1 = kProfile ? (long long)__builtin_readcyclecounter() : 0;
    if (upd) {
        // ---- daqp_update_ldp(UPDATE_v|UPDATE_d) on the resident factors (utils.c:58-221 without Rinv/M), cf. k_update
        // check_bounds (utils.c:546-567) on the stored sense: the first crossed pair (in index order) ends the update with
        // -1; unmarked equalities before it have been marked by then, nothing else changes (see k_update)
        int first_bad = kBig;
        static_for<NB>([&](auto bb) __attribute__((always_inline)) {
            const int r = bb * 64 + lane;
            if (r < m && !(rsense_get(w, bb) & DAQP_IMMUTABLE) && bur[bb] - blr[bb] < -w.stp->primal_tol && first_bad == kBig) first_bad = r;
        });
        first_bad = (int)wave_min((double)first_bad);
        int bad = 0;
        static_for<NB>([&](auto bb) __attribute__((always_inline)) {
            const int r = bb * 64 + lane;
            const int sn = rsense_get(w, bb);
            if (r < m && r < first_bad && !(sn & DAQP_IMMUTABLE) && !(sn & DAQP_SOFT) && bur[bb] - blr[bb] < w.stp->zero_tol) {
                w.rs |= (typename RWave<NB, NP, FM, IMG>::rs_t)(DAQP_ACTIVE | DAQP_IMMUTABLE) << (8 * bb); bad |= 4;
            }
        });
        if (first_bad != kBig) {
            static_for<NB>([&](auto bb) __attribute__((always_inline)) { const int r = bb * 64 + lane; if (r < m) gsense[r] = rsense_get(w, bb); });
            if (lane == 0) {
                qs->upd_flag = DAQP_EXIT_INFEASIBLE; qs->sing_ind = kEmpty;
                as_global(b.exitflag)[q] = DAQP_EXIT_INFEASIBLE; as_global(b.iter)[q] = 0; if (b.fval) as_global(b.fval)[q] = 0; if (b.soft) as_global(b.soft)[q] = 0;
            }
            copy_wait();   // nothing may still be landing in LDS when the workgroup ends
            return;
        }
        if (lane == 0) qs->upd_flag = 0;
        if (__any(bad & 4)) q_need_act = 1;
        double *vv = w.u, *fl = w.pend_lam;       // both regions are free until the loop starts (u is zeroed below)
        for (int e = lane; e < round_up(n > 64 ? n : 64, 2) + 2; e += 64) vv[e] = 0;
        if (upd & DAQP_UPDATE_v) {   // v = R^-T f (utils.c:474-497): formed between the row loads above, or here
            double acc = v_early;
            if (!v_done) {
                if (lane < n) fl[lane] = f_in;
                copy_wait();
                WSYNC();
                acc = v_of_update(Rl0, fl);
            }
            if (lane < n) {
                vv[lane] = acc;
                as_global(b.v)[(size_t)q * n + lane] = acc;
            }
        } else if (lane < n) vv[lane] = as_global(b.v)[(size_t)q * n + lane];
        WSYNC();
        // d = b*scaling + (row . v) for every row of the dense image (utils.c:499-544); rows stay in registers
        {
            const double2 *v2 = reinterpret_cast<const double2 *>(vv);
            double sm[NB];
            static_for<NB>([&](auto bb) __attribute__((always_inline)) { sm[bb] = sm_img[bb]; });
            if constexpr (IMG == 0)
            static_for<NP>([&](auto tt) __attribute__((always_inline)) {
                const double2 vk = v2[tt];
                static_for<NB>([&](auto bb) __attribute__((always_inline)) {
                    sm[bb] = madd<FM>(sm[bb], w.Mx[bb][tt], vk.x);
                    sm[bb] = madd<FM>(sm[bb], w.My[bb][tt], vk.y);     // odd n: the last pair's partner is 0 * 0 (both zero padding), an exact no-op
                });
            });
            static_for<NB>([&](auto bb) __attribute__((always_inline)) {
                const int r = bb * 64 + lane;
                if (r < m) {
                    const double nu = bur[bb] * scr[bb] + sm[bb], nl = blr[bb] * scr[bb] + sm[bb];
                    dur[bb] = nu; dlr[bb] = nl;
                    w.rowv[r] = nu; w.rowv[kRowvStride<NB, IMG> + r] = nl;
                    as_global(b.dupper)[(size_t)q * m + r] = nu;
                    as_global(b.dlower)[(size_t)q * m + r] = nl;
                }
            });
        }
        w.reuse = 0; w.sing = kEmpty;     // utils.c:80-81
        WSYNC();
    }
    // ---- working-set view
    w.wsid = wsid_r;
    w.slot = lane;
    w.slotmask = (w.na >= 64) ? ~0ull : ((1ull << w.na) - 1ull);
    w.hi_slot = w.na - 1;
    const int rinv_off = rowc_size - round_up(b.rtri, 2) - 2;
    w.D = D_r; w.xl = xl_r; w.zl = zl_r;
    w.lam = swapped ? lb_r : la_r;
    w.lams = swapped ? la_r : lb_r;
    {   // sense and bound of each working-set row are already in registers (row view): fetch them across lanes
        // (ds_bpermute) instead of a second, dependent trip to HBM
        const int src = w.wsid & 63, blk = w.wsid >> 6;
        int fl = 0;
        double bu = 0, bl = 0;
        static_for<NB>([&](auto bb) __attribute__((always_inline)) {
            int s8;
            if constexpr (NB > 4) s8 = (int)((__shfl((long long)w.rs, src) >> (8 * bb)) & 0xff);
            else s8 = (int)((__shfl((int)w.rs, src) >> (8 * bb)) & 0xff);
            const double u_ = __shfl(dur[bb], src), l_ = __shfl(dlr[bb], src);
            if (blk == bb) { fl = s8; bu = u_; bl = l_; }
        });
        w.wflag = act ? fl : 0;
        w.drhs = act ? -((fl & DAQP_LOWER) ? bl : bu) : 0.0;
    }
    {
        // warm start: packed L and the active-row cache come straight from HBM into LDS (global_load_lds, no VGPRs),
        // every instruction in flight at once -- one memory round trip instead of one per row
        for (int e = lane; e < round_up(n > 64 ? n : 64, 2) + 2; e += 64) w.u[e] = 0;
        if (!rows_early) fetch_active_rows();
        if (kProfile && w.prof && lane == 0) { w.prof[26] = tp1 - t_start; w.prof[27] = (long long)__builtin_readcyclecounter() - tp1; }
        copy_wait();
        if (kProfile && w.prof && lane == 0) w.prof[31] = (long long)__builtin_readcyclecounter() - tp1;
    }
    WSYNC();

    int iters = 0;
    const long long t_loop = (long long)__builtin_readcyclecounter();
    int flag = rrun(w, mode, q_need_act != 0, iters);
    const long long t_done = (long long)__builtin_readcyclecounter();
    if constexpr (kRegHandOver<NB, NP, IMG>) {
        // nothing of this problem has been stored yet (results, iterate, sense, record: all below): the generic kernel starts from the same state
        if (flag == kRegHandOverFlag) {
            if (b.fallback != nullptr) {
                if (lane == 0) { as_global(b.fallback)[q] = 1; if (IMG != 0 && b.img_ho) atomicAdd(b.img_ho, 1); }
                copy_wait();
                return;
            }
            flag = DAQP_EXIT_UNSUPPORTED;      // (nobody stands behind this launch: cannot happen with the host code of this library)
        }
    }
    if (mode == 1) {
        if (lane == 0) { qs->need_activate = 0; if (flag < 0) { qs->setup_flag = flag; qs->exitflag = flag; } }
    } else {
        const double *Rq = b.Rinv + qfac * b.rtri;
        // Everything that needs LDS hand-offs (x from the staged R^-1, lam assembled by working-set scatter) comes
        // first; the stores to HBM are issued together at the very end, so no fence ever waits for a store.
        // Packed R^-1 comes into the top of the (now dead) active-row cache in one HBM round trip; v and the scalings ride along.
        // (16 bytes per lane; the packed rows of odd QPs start 8 bytes off a 16-byte boundary, so the LDS image is
        // shifted by one double for them and that first double goes separately)
        const int odd8 = (int)(((size_t)Rq >> 3) & 1);
        if constexpr (IMG != 0) {      // the staging area starts at L: the factor goes to HBM first
            const int used = tri(w.na);
            DAQP_GLOBAL(double) *gL = as_global(b.L + (size_t)q * b.ltri);
            for (int e = lane; e < used; e += 64) gL[e] = w.L[e];
            WSYNC();
        }
        double *Rl = sbase + rinv_off + odd8;
        if (flag > 0) {
            if (odd8) copy_async_dwords(Rl, Rq, 1);
            const int body = (b.rtri - odd8) & ~1;
            copy_async(Rl + odd8, Rq + odd8, body);
            if (odd8 + body < b.rtri) copy_async_dwords(Rl + odd8 + body, Rq + odd8 + body, 1);
        }
        const DAQP_GLOBAL(double) *vq = as_global(b.v + (size_t)q * n);
        const double vl = (lane < n) ? vq[lane] : 0.0;
        const double sc_ws = (flag > 0 && lane < w.na) ? gsc[w.wsid] : 1.0;
        const double sc_sb = (flag > 0 && lane < b.ms) ? gsc[lane] : 1.0;
        double *lamq = sbase;                                           // m doubles at the (dead) bottom of the row cache
        double xi = (lane < n) ? w.u[lane] : 0.0;
        if (b.lam) for (int i = lane; i < m; i += 64) lamq[i] = 0;     // daqp_extract_result (api.c:455-495): zero ...
        const long long te1 = (long long)__builtin_readcyclecounter();
        copy_wait();
        if (flag > 0) {   // ldp2qp_solution (daqp.c:111-139)
            if (lane < n) w.u[lane] = w.u[lane] - vl;
            if (lane < w.na) w.lams *= sc_ws;
        }
        WSYNC();
        const long long te2 = (long long)__builtin_readcyclecounter();
        if (b.lam && lane < w.na) lamq[w.wsid] = w.lams;                // ... then scatter by WS
        if constexpr (FM) {
            // x = R^-1 (u - v), default arithmetic: the triangle folded as for v above -- row i has n - i terms; lane i takes the first
            // LH of its own row (j = i, i+1, ...) and the tail of row n-1-i that its owner leaves over (j = n-1-i+LH .. n-1)
            if (flag > 0) {      // (wave-uniform)
                const int i = lane < n ? lane : 0, pr = n - 1 - i;
                const int LH = n / 2 + 1;
                const int own_hi = (n - i > LH) ? i + LH - 1 : n - 1;                  // own terms: j = i .. own_hi
                const int tail_lo = (n - pr > LH && pr != i) ? pr + LH : n;            // partner's left-over: j = tail_lo .. n-1
                const double *row = Rl + roff(i, n), *prow = Rl + roff(pr, n);
                double acc = 0, acc2 = 0;
                for (int j0 = i; j0 <= own_hi; j0 += kChunk) {
                    double rr[kChunk], uu[kChunk];
#pragma unroll
                    for (int k = 0; k < kChunk; ++k) { const int j = (j0 + k <= own_hi) ? j0 + k : own_hi; rr[k] = row[j]; uu[k] = w.u[j]; }
#pragma unroll
                    for (int k = 0; k < kChunk; ++k) if (j0 + k <= own_hi) acc = __builtin_fma(rr[k], uu[k], acc);
                }
                for (int j0 = tail_lo; j0 < n; j0 += kChunk) {
                    double rr[kChunk], uu[kChunk];
#pragma unroll
                    for (int k = 0; k < kChunk; ++k) { const int j = (j0 + k < n) ? j0 + k : n - 1; rr[k] = prow[j]; uu[k] = w.u[j]; }
#pragma unroll
                    for (int k = 0; k < kChunk; ++k) if (j0 + k < n) acc2 = __builtin_fma(rr[k], uu[k], acc2);
                }
                const double other = __shfl(acc2, (lane < n) ? n - 1 - lane : lane);
                if (lane < n) {
                    xi = acc + other;
                    if (lane < b.ms && !qdiag) xi /= sc_sb;   // daqp.c:124-134: no division in the RinvD branch
                }
            }
        } else if (flag > 0 && lane < n) {
            const double *row = Rl + roff(lane, n);
            xi = w.u[lane] * row[lane];
            for (int j0 = lane + 1; j0 < n; j0 += kChunk) {
                double rr[kChunk], uu[kChunk];
#pragma unroll
                for (int k = 0; k < kChunk; ++k) { const int j = (j0 + k < n) ? j0 + k : n - 1; rr[k] = row[j]; uu[k] = w.u[j]; }
#pragma unroll
                for (int k = 0; k < kChunk; ++k) if (j0 + k < n) xi += rr[k] * uu[k];
            }
            if (lane < b.ms && !qdiag) xi /= sc_sb;   // daqp.c:124-134: no division in the RinvD branch
        }
        WSYNC();
        const long long te3 = (long long)__builtin_readcyclecounter();
        if (kProfile && w.prof && lane == 0) { w.prof[20] = te1 - t_done; w.prof[21] = te2 - te1; w.prof[22] = te3 - te2; }
        if (b.x && lane < n) as_global(b.x)[(size_t)q * n + lane] = xi;
        if (b.lam) for (int i = lane; i < m; i += 64) as_global(b.lam)[(size_t)q * m + i] = lamq[i];
        double fv = w.fval;                      // fval - |v|^2: in index order, v_i broadcast from its lane -- or, default mode, by tree
        if constexpr (FM) fv -= wave_sum(vl * vl);
        else static_for<8>([&](auto c) __attribute__((always_inline)) {
            if (8 * c < n) static_for<8>([&](auto k) __attribute__((always_inline)) { const double vi = rl(vl, 8 * c + k); fv -= vi * vi; });
        });
        fv *= 0.5;
        if (lane == 0) {
            as_global(b.exitflag)[q] = flag; as_global(b.iter)[q] = iters;
            if (b.fval) as_global(b.fval)[q] = fv;
            if (b.soft) as_global(b.soft)[q] = w.soft;
            qs->iterations = iters; qs->exitflag = flag; qs->need_activate = 0;
        }
    }
    // ---- store the persistent iterate (lam in buffer A, lam* in buffer B)
    if (lane < cap) {
        gv[lane] = w.D; gv[cap + lane] = w.xl; gv[2 * cap + lane] = w.zl;
        gv[3 * cap + lane] = w.lam; gv[4 * cap + lane] = w.lams;
        gws[lane] = (lane < w.na) ? w.wsid : -1;
    }
    static_for<NB>([&](auto bb) __attribute__((always_inline)) { const int r = bb * 64 + lane; if (r < m) gsense[r] = rsense_get(w, bb); });
    if (IMG == 0 || mode == 1) {
        const int used = tri(w.na);
        DAQP_GLOBAL(double) *gL = as_global(b.L + (size_t)q * b.ltri);
        for (int e = lane; e < used; e += 64) gL[e] = w.L[e];
    }
    if (lane == 0) {
        qs->n_active = w.na; qs->reuse_ind = w.reuse; qs->sing_ind = w.sing;
        qs->lam_swapped = 0;
        qs->fval = w.fval; qs->soft_slack = w.soft;
        if (b.trace) as_global(b.trace)[(size_t)q * b.trace_cap + b.trace_cap - 1] = w.trace_len;
        if (kProfile && w.prof) {
            const long long te4 = (long long)__builtin_readcyclecounter();
            w.prof[23] = te4 - t_done;
            for (int i = 0; i < 32; ++i) as_global(b.prof)[(size_t)q * 32 + i] = w.prof[i];
            as_global(b.prof)[(size_t)q * 32 + 28] = t_loop - t_start;                                    // prologue
            as_global(b.prof)[(size_t)q * 32 + 29] = (long long)__builtin_readcyclecounter() - t_done;   // epilogue
            as_global(b.prof)[(size_t)q * 32 + 30] = t_done - t_loop;                                     // the loop
        }
    }
}

} // namespace daqp_amd
