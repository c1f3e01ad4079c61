// kernels.hip.h -- the three device entry points of the path, one wavefront per QP:
//   k_setup   QP -> LDP        (reference src/utils.c:58-687: check_bounds, Cholesky, R^-1, v,
//                               unconstrained shortcut, M = A R^-1, normalisation, d)
//   k_update  new f / bounds   (reference daqp_update_ldp with DAQP_UPDATE_v | DAQP_UPDATE_d)
//   k_ldp     solve            (reference daqp_solve: daqp_ldp, ldp2qp_solution, daqp_extract_result)
#pragma once
#include "wave_ldp.hip.h"
#include "wave_ldp_reg.hip.h"
#include "batch_dev.hip.h"

namespace daqp_amd {

// ------------------------------------------------------------------------------------
// LDS carve-ups (doubles first, ints last; every offset a multiple of 16 bytes)
// ------------------------------------------------------------------------------------
// rows of A per pass of k_setup's M phase (LDS: this many rows of A plus as many rows of one 64-column result block)
#ifndef DAQP_AMD_SETUP_ROWS
#define DAQP_AMD_SETUP_ROWS 4
#endif
constexpr int kSetupRows = DAQP_AMD_SETUP_ROWS;
#ifndef DAQP_AMD_CHOL_DEPTH
#define DAQP_AMD_CHOL_DEPTH 4
#endif
struct SetupLds { int R, Rout, fv, vv, xu, sc, du, dl, tile, sens, total_bytes; };
__host__ __device__ inline SetupLds setup_lds(int n, int m, bool gs = false)
{
    SetupLds s;
    const int rt = round_up(n * (n + 1) / 2, 2), np = round_up(n, 2);
    int o = 0;
    s.R = o; if (!gs) o += rt; s.Rout = o; if (!gs) o += rt;
    s.fv = o; o += np; s.vv = o; o += np; s.xu = o; o += np;
    s.sc = s.du = s.dl = o;   // (not in LDS any more)
    {   // the block of rows of A and the result block of the exact M phase; the matrix-core phase needs 16 x 64 results
        const int t = kSetupRows * np + kSetupRows * 64;
        s.tile = o; o += t > 1024 ? t : 1024;
    }
    s.sens = o;
    s.total_bytes = o * 8 + round_up(m, 4) * 4;
    return s;
}
struct LdpLds { int L, rowc, rowc_size, D, xl, zl, lamA, lamB, u, pend_lam, rowv, dbl; int ws, sense, pend_id, ints; int total_bytes; };
__host__ __device__ inline LdpLds ldp_lds(int n, int m, int cap, bool spill, int ldrc = 0)
{
    LdpLds s;
    const int ldr = ldrc > 0 ? ldrc : (n | 1), cp = round_up(cap, 2);
    int o = 0;
    s.L = o; if (!spill) o += round_up(cap * (cap + 1) / 2, 2);
    s.rowc = o;
    if (!spill) {
        int rc = round_up(cap * ldr, 2);
        const int fin = round_up(n * (n + 1) / 2, 2) + 2 + round_up(m, 2);   // register kernel's epilogue: staged R^-1 + lam
        if (ldrc > 0 && rc < fin) rc = fin;
        s.rowc_size = rc;
        o += rc;
    } else s.rowc_size = 0;
    s.D = o; o += cp; s.xl = o; o += cp; s.zl = o; o += cp; s.lamA = o; o += cp; s.lamB = o; o += cp;
    s.u = o; o += round_up(n > 64 ? n : 64, 2) + 2;   // zero-padded to 64 for the register-resident scan
    s.pend_lam = o; o += cp;
    s.rowv = o;
    s.dbl = o;                       // ints start at double offset s.dbl
    int oi = 0;
    s.ws = oi; oi += round_up(cap, 4);
    s.sense = oi; oi += round_up(m, 4);
    s.pend_id = oi; oi += round_up(cap, 4);
    s.ints = oi;
    s.total_bytes = o * 8 + oi * 4;
    return s;
}

// ------------------------------------------------------------------------------------
// k_setup: one wave per QP
// ------------------------------------------------------------------------------------
// GS = true: the packed Cholesky factor and R^-1 live in per-QP HBM scratch (b.setup_g)
// instead of LDS -- the n = 200 class of problems, where 2 x 160 KB of factors cannot be staged.
// NBK = 64-column blocks a lane covers (one entry per block and lane): 4 for n <= 256; 8 (n <= 512: the reference's own "large"
// benchmark ladder goes to n = 500, interfaces/daqp-julia/test/benchmark.jl:38) with half as many rows per group, so that
// the register footprint of the Cholesky / inverse sweeps stays the same.
#ifndef DAQP_AMD_SETUP_DEFER_WAVES
#define DAQP_AMD_SETUP_DEFER_WAVES 2
#endif
// DEFER: the instantiation that is followed by k_setup_m (setup_m.hip.h) -- the general rows are compiled out, and with them the A
// operand of the matrix-core phase.  What is left (Cholesky, inverse, v, simple bounds) is latency-bound per wave, but it does NOT
// fit the register budget of a third wave per SIMD: at 168 registers the sweeps' row groups spill (125 registers, 196 bytes of
// scratch) and the launch takes 2.3x as long (measured on config C4: 39.0 ms against 17.0 per 4 096 problems) -- two waves it is
// PART: daqp_update_ldp with a mask that is neither within v|d (k_update) nor everything (utils.c:58-221 bit by bit; the masks
// the reference's own bindings send field by field, daqp.pyx:513-571).  The instantiation of its own keeps these branches out of
// the setup launches the configurations are timed on.  It always runs in the reference's operation order (host: exact_setup = 1).
//   sense bit clear   -> the workspace's sense is KEPT as the last solve / update left it, stale ACTIVE bits included (utils.c:84-91)
//   no M, v or d bit  -> no bound check (utils.c:94-98); with one: marks up to the first crossed pair, which ends the update (-1)
//   no Rinv bit       -> R^-1 as stored (rows < ms normalised): new f and new A are divided by those rows' scalings column by column
//                        (utils.c:447-452, 491-496), R^-1 is not normalised again, d is formed anew from whatever changed
//   Rinv bit          -> everything below as in a setup (v, M, normalisation, d follow the new factor: utils.c:122,135,142,150)
//   the working set is reset (daqp_update_M, utils.c:470) and re-activated only when sense was given or an equality was marked
template <bool GS, int NBK = 4, bool DEFER = false, bool PART = false>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(DEFER ? DAQP_AMD_SETUP_DEFER_WAVES : 2, DEFER ? DAQP_AMD_SETUP_DEFER_WAVES : 2))) void k_setup(BatchDev b, int mask)
{
    static_assert(!(PART && DEFER), "partial updates form their general rows themselves");
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int q = blockIdx.x, lane = lane_id();
    const int n = b.n, m = b.m, ms = b.ms, mA = b.mA, ldr = b.ldr;
    const SetupLds o = setup_lds(n, m, GS);
    double *R, *Ro;
    if constexpr (GS) {
        const size_t per = 2 * (size_t)round_up(b.rtri, 2);
        double *g = b.setup_g + (size_t)q * per;
        R = g; Ro = g + round_up(b.rtri, 2);
    } else { R = smem + o.R; Ro = smem + o.Rout; }
    double *fl = smem + o.fv, *vv = smem + o.vv, *xu = smem + o.xu;
    // scaling and d go straight to their HBM arrays (LDS per workgroup decides how many problems share a CU here)
    double *sc = b.scaling + (size_t)q * m, *du = b.dupper + (size_t)q * m, *dl = b.dlower + (size_t)q * m;
    int *sens = reinterpret_cast<int *>(smem + o.sens);
    const double *H = b.H + (b.prox_pass == 2 ? (size_t)0 : (size_t)q * n * n), *f = b.f + (size_t)q * n, *A = b.A + (size_t)q * mA * n;
    const double *bu = b.bu + (size_t)q * m, *bl = b.bl + (size_t)q * m;
    const DAQPSettings &st = b.st;
    QState *qs = b.qs + q;
    int flag = 1, activate = 0;
    const bool doR = !PART || (mask & DAQP_UPDATE_Rinv) != 0;                       // new Hessian: factorise
    const bool doV = doR || (mask & DAQP_UPDATE_v) != 0;                            // v follows R^-1 or f
    const bool newS = !PART || (mask & DAQP_UPDATE_sense) != 0;
    const bool chk = !PART || (mask & (DAQP_UPDATE_M | DAQP_UPDATE_v | DAQP_UPDATE_d)) != 0;
    const bool keepR = PART && !doR;
    if constexpr (PART) { if (keepR && __builtin_amdgcn_readfirstlane(qs->setup_flag) < 0) return; }   // no factor to update on
    // a regularising re-run (utils.c:356-377): the flagged problems only, H + shift*I (or, for a diagonal H, the shift in
    // its singular coordinates only: utils.c:284-312), the stricter pivot ratio for a Hessian that needed the shift
    const int pp = b.prox_pass;
    if (pp && __builtin_amdgcn_readfirstlane(qs->setup_flag) != DAQP_NEEDS_SHIFT) return;
    const double shift = pp ? b.hshift[q] : 0.0;
    const bool force = st.eps_prox > 0.0 && pp != 2;
    const int shift_code = (!pp && st.eps_prox == 0.0) ? DAQP_EXIT_NONCONVEX : DAQP_NEEDS_SHIFT;   // eps == 0: utils.c:357,367
    int nprox = 0;
    // default arithmetic mode: M = A R^-1 on the matrix cores (below), which read R^-1 from a zero-padded square image
    bool mfma_m = !PART && !b.exact_setup && b.setup_sq != nullptr && b.n <= 208;   // (52 k steps of A in registers)
    // ... and, when the host follows this launch with k_setup_m (setup_m.hip.h: a workgroup per 64 rows of A, R^-1 shared through
    // LDS), the general rows are not formed here at all: this kernel leaves what is owed in qs->pad_
    const int sq_ld = round_up(n, 16);
    double *Rsq = b.setup_sq ? b.setup_sq + (size_t)q * round_up(n, 32) * sq_ld : nullptr;
    // optional phase cycle counters -> b.prof[q][16..21]: checks, Cholesky, inverse, v/x_unc, M rows, simple bounds + write-back
    long long gpt[6] = {0, 0, 0, 0, 0, 0};
    long long gt0 = b.prof ? (long long)__builtin_readcyclecounter() : 0;
#define GPROF(slot) do { if (b.prof) { const long long t1 = (long long)__builtin_readcyclecounter(); gpt[slot] += t1 - gt0; gt0 = t1; } } while (0)

    // --- sense (utils.c:84-91) and early bound check (utils.c:546-567)
    int bad = 0;
    if constexpr (!PART) {
    for (int i = lane; i < m; i += 64) {
        int s = b.sense_in ? b.sense_in[(size_t)q * m + i] : 0;
        if (s & DAQP_BINARY) bad |= 2;
        if (!(s & DAQP_IMMUTABLE)) {
            const double diff = bu[i] - bl[i];
            if (diff < -st.primal_tol) bad |= 1;
            else if (diff < st.zero_tol && !(s & DAQP_SOFT)) { s |= DAQP_ACTIVE + DAQP_IMMUTABLE; bad |= 4; }
        }
        sens[i] = s;
    }
    {
        const int any = __any(bad & 2) ? 2 : 0, inf = __any(bad & 1) ? 1 : 0, eq = __any(bad & 4) ? 4 : 0;
        bad = any | inf | eq;
    }
    } else {
        // the kept sense, or the caller's; a regularising re-run (pp) continues from what its first pass left (sense copied, bounds checked)
        int *cur = b.sense + (size_t)q * m;
        const bool from_ws = !newS || pp != 0, check = chk && !pp;
        int first_bad = kBig;
        for (int i = lane; i < m; i += 64) {
            const int s = from_ws ? cur[i] : (b.sense_in ? b.sense_in[(size_t)q * m + i] : 0);
            if (s & DAQP_BINARY) bad |= 2;
            sens[i] = s;
            if (check && !(s & DAQP_IMMUTABLE) && bu[i] - bl[i] < -st.primal_tol && first_bad == kBig) first_bad = i;
        }
        first_bad = (int)wave_min((double)first_bad);      // the reference walks the rows in order and returns at the first crossed pair:
        if (check)                                         // unmarked equalities BEFORE it have been marked by then (k_update)
            for (int i = lane; i < m && i < first_bad; i += 64) {
                const int s = sens[i];
                if (!(s & DAQP_IMMUTABLE) && !(s & DAQP_SOFT) && bu[i] - bl[i] < st.zero_tol) { sens[i] = s | DAQP_ACTIVE | DAQP_IMMUTABLE; bad |= 4; }
            }
        bad = (__any(bad & 2) ? 2 : 0) | (__any(bad & 4) ? 4 : 0);
        if (bad & 2) {                                     // binary constraints are outside this path: nothing is taken over
            if (lane == 0) qs->upd_flag = DAQP_EXIT_UNSUPPORTED;
            return;
        }
        if (first_bad != kBig) {                           // utils.c:95-96: the update ends here, the factors are untouched
            for (int i = lane; i < m; i += 64) cur[i] = sens[i];
            if (lane == 0) { qs->upd_flag = DAQP_EXIT_INFEASIBLE; qs->sing_ind = kEmpty; qs->need_activate = 0; }
            return;
        }
    }
    if constexpr (PART) {
        if (pp) activate = __builtin_amdgcn_readfirstlane(qs->need_activate);   // (what the first pass decided)
        else if (newS && b.sense_in) activate = 1;
    } else if (b.sense_in) activate = 1;
    if (bad & 4) activate = 1;
    if (bad & 2) flag = DAQP_EXIT_UNSUPPORTED;
    else if (bad & 1) flag = DAQP_EXIT_INFEASIBLE;
    if (force && !pp && flag > 0 && doR) flag = DAQP_NEEDS_SHIFT;   // forced proximal mode (utils.c:233-281): the host starts with the shifted pass
    const int rec_diag = keepR ? __builtin_amdgcn_readfirstlane(qs->diag_h) : 0;
    if (doV) for (int i = lane; i < n; i += 64) fl[i] = (keepR && i < ms && !rec_diag) ? f[i] / sc[i] : f[i];   // utils.c:491-496: rows < ms of the kept R^-1 are normalised
    WSYNC();

    // --- Cholesky of 1/2(H+H') in packed-upper form, 1/r_ii on the diagonal (utils.c:318-352).
    // Row i: lane <-> column j, k-ordered subtraction chain kept in a register.
    double pmin = DAQP_INF, pmax = 0.0;
    int diag = 0;   // H diagonal: the reference's RinvD branch (utils.c:245-312), as in k_setup_fast
    // default arithmetic, 64 < n <= 200: k_fact_wg (setup_fact.hip.h: a workgroup per problem, the triangle in LDS, matrix cores) has
    // factored and inverted already -- R^-1 sits in the scratch and in the square image, the record carries the pivots' range
    bool factored = false;
    if constexpr (GS) {
        if (b.fact != nullptr && !pp) {
            const double *rec = b.fact + (size_t)q * 4;
            factored = __builtin_amdgcn_readfirstlane((int)rec[0]) != 0;
            if (factored) { pmin = rec[1]; pmax = rec[2]; }
        }
    }
    if constexpr (PART) if (keepR) {   // the stored factor (rows < ms normalised), its kind and its shift
        diag = rec_diag; nprox = __builtin_amdgcn_readfirstlane(qs->n_prox);
        for (int e = lane; e < b.rtri; e += 64) Ro[e] = b.Rinv[(size_t)q * b.rtri + e];
        WSYNC();
    }
    if (flag > 0 && !factored && doR) {
        int offd = 0;
        for (int e = lane; e < n * n; e += 64) {
            const int i = e / n, j = e - i * n;
            if (j > i && (H[e] > st.zero_tol || H[e] < -st.zero_tol)) offd = 1;   // entries above the diagonal only (utils.c:245-252)
        }
        if (!__any(offd)) {
            double hmax = 0;
            for (int i = lane; i < n; i += 64) { double a_ = H[(size_t)i * n + i]; if (a_ < 0) a_ = -a_; if (a_ > hmax) hmax = a_; }
            const double hscale = -wave_min(-hmax);
            const double ftol = hscale > 0 ? st.zero_tol * hscale : st.zero_tol;
            for (int e = lane; e < b.rtri; e += 64) Ro[e] = 0.0;
            WSYNC();
            for (int ic = 0; ic < n && flag > 0; ic += 64) {   // ascending: the reference stops at the first bad entry
                const int i = ic + lane;
                double hd = (i < n) ? H[(size_t)i * n + i] : 1.0;
                const bool low = i < n && (hd <= ftol || (pp && force));
                bool fail = low;
                int code = (st.eps_prox == 0.0 && hd <= st.zero_tol) ? DAQP_EXIT_NONCONVEX : DAQP_NEEDS_SHIFT;
                if (pp) {   // semi-proximal: shift the singular coordinates only, remember which (utils.c:294-303)
                    if (low) hd += shift;
                    if (i < n) b.prox_mask[(size_t)q * n + i] = low ? 1 : 0;
                    nprox += __popcll(__ballot(low));
                    fail = i < n && hd <= st.zero_tol;
                    code = DAQP_EXIT_NONCONVEX;
                    if (pp == 2 && i < n) b.prox_mask[(size_t)q * n + i] = 1;   // LP (api.c:183-185): n_prox = n
                }
                const unsigned long long fm = __ballot(fail);
                if (fm) { flag = __builtin_amdgcn_readlane(code, __ffsll((long long)fm) - 1); break; }
                if (i < n) {
                    const double hsq = sqrt(hd);
                    Ro[roff(i, n) + i] = 1 / hsq;
                    if (i < ms) sc[i] = hsq;
                }
            }
            diag = 1;
            mfma_m = false;   // (M = A D^-1/2: the chain below; the square image is not written on this path)
            if (pp == 2) nprox = n;
            WSYNC();
        }
    }
    GPROF(0);
    if (flag > 0 && factored && pmin <= st.zero_tol * pmax) flag = shift_code;   // utils.c:354-356
    if (flag > 0 && !diag && !factored && doR) {
        for (int e0 = lane; e0 < n * n; e0 += 64 * 8) {   // 16 loads per lane per trip (H and its transpose), then the stores
            double h1[8], h2[8];
            int ii[8], jj[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int e = (e0 + 64 * u < n * n) ? e0 + 64 * u : 0;
                ii[u] = e / n; jj[u] = e - ii[u] * n;
                h1[u] = H[e]; h2[u] = H[(size_t)jj[u] * n + ii[u]];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (e0 + 64 * u < n * n && jj[u] >= ii[u]) R[roff(ii[u], n) + jj[u]] = (ii[u] == jj[u]) ? h1[u] + shift : 0.5 * (h1[u] + h2[u]);
        }
        WSYNC();
        // Left-looking, KR rows at a time, lane <-> column (one entry per 64-column block and lane): the finished rows k < i0
        // are streamed once for the whole group (a coalesced stretch per block plus the KR entries R[k][i0..] as broadcasts),
        // then the rows of the group finish one after the other, each updating the later ones from registers.  Every entry
        // receives its subtractions in ascending k (utils.c:335-352).
        {
            constexpr int KR = 32 / NBK;
            for (int i0 = 0; i0 < n && flag > 0; i0 += KR) {
                const int b0 = i0 >> 6;
                double acc[KR][NBK];
                static_for<KR>([&](auto rr) __attribute__((always_inline)) {
                    constexpr int r = rr;
                    const int i = i0 + r;
                    const int pi = roff(i < n ? i : 0, n);
                    static_for<NBK>([&](auto jj) __attribute__((always_inline)) {
                        constexpr int jb = jj;
                        const int j = jb * 64 + lane;
                        acc[r][jb] = (i < n && j >= i && j < n) ? R[pi + j] : 0.0;
                    });
                });
                constexpr int KD = DAQP_AMD_CHOL_DEPTH;   // finished rows in flight (i0 is a multiple of KR, KR of KD)
                for (int k0 = 0; k0 < i0; k0 += KD) {
                    double rk[KD][NBK], ru[KD][KR];
                    static_for<KD>([&](auto tt) __attribute__((always_inline)) {
                        constexpr int t = tt;
                        const int pk = roff(k0 + t, n);
                        static_for<NBK>([&](auto jj) __attribute__((always_inline)) {
                            constexpr int jb = jj;
                            const int j = jb * 64 + lane;
                            rk[t][jb] = (jb >= b0 && j >= i0 && j < n) ? R[pk + j] : 0.0;
                        });
                        static_for<KR>([&](auto rr) __attribute__((always_inline)) {
                            constexpr int r = rr;
                            ru[t][r] = R[pk + ((i0 + r < n) ? i0 + r : i0)];
                        });
                    });
                    static_for<KD>([&](auto tt) __attribute__((always_inline)) {
                        constexpr int t = tt;
                        static_for<KR>([&](auto rr) __attribute__((always_inline)) {
                            constexpr int r = rr;
                            static_for<NBK>([&](auto jj) __attribute__((always_inline)) {
                                constexpr int jb = jj;
                                if (jb >= b0) acc[r][jb] -= ru[t][r] * rk[t][jb];
                            });
                        });
                    });
                }
                static_for<KR>([&](auto rr) __attribute__((always_inline)) {
                    constexpr int r = rr;
                    const int i = i0 + r;
                    if (i < n && flag > 0) {
                        const int bi = i >> 6, li = i & 63;
                        double dsel = acc[r][0];
                        if (bi == 1) dsel = acc[r][1];
                        if (bi == 2) dsel = acc[r][2];
                        if (bi == 3) dsel = acc[r][3];
                        if constexpr (NBK == 8) {   // (plain selects: through a lambda the array was not promoted to registers -- 260 bytes of scratch, C4 setup 47 -> 119 ms)
                            if (bi == 4) dsel = acc[r][4];
                            if (bi == 5) dsel = acc[r][5];
                            if (bi == 6) dsel = acc[r][6];
                            if (bi == 7) dsel = acc[r][7];
                        }
                        const double dg = rl(dsel, li);
                        if (dg <= st.zero_tol) flag = shift_code;
                        else {
                            if (dg < pmin) pmin = dg;
                            if (dg > pmax) pmax = dg;
                            const double dgi = 1 / sqrt(dg);
                            static_for<NBK>([&](auto jj) __attribute__((always_inline)) {
                                constexpr int jb = jj;
                                const int j = jb * 64 + lane;
                                acc[r][jb] = (j == i) ? dgi : acc[r][jb] * dgi;
                            });
                            const int pi = roff(i, n);
                            static_for<NBK>([&](auto jj) __attribute__((always_inline)) {
                                constexpr int jb = jj;
                                const int j = jb * 64 + lane;
                                if (j >= i && j < n) R[pi + j] = acc[r][jb];
                            });
                            // this row is the next k for the later rows of the group
                            static_for<KR>([&](auto qq) __attribute__((always_inline)) {
                                constexpr int r2 = qq;
                                if constexpr (r2 > r) {
                                    const int i2 = i0 + r2;
                                    if (i2 < n) {
                                        const int b2 = i2 >> 6, l2 = i2 & 63;
                                        double ssel = acc[r][0];
                                        if (b2 == 1) ssel = acc[r][1];
                                        if (b2 == 2) ssel = acc[r][2];
                                        if (b2 == 3) ssel = acc[r][3];
                                        if constexpr (NBK == 8) {
                                            if (b2 == 4) ssel = acc[r][4];
                                            if (b2 == 5) ssel = acc[r][5];
                                            if (b2 == 6) ssel = acc[r][6];
                                            if (b2 == 7) ssel = acc[r][7];
                                        }
                                        const double sv = rl(ssel, l2);
                                        static_for<NBK>([&](auto jj) __attribute__((always_inline)) {
                                            constexpr int jb = jj;
                                            if (jb >= b0) acc[r2][jb] -= sv * acc[r][jb];
                                        });
                                    }
                                }
                            });
                        }
                    }
                });
                WSYNC();
            }
        }
        if (flag > 0 && pmin <= ((pp && !force) ? sqrt(st.zero_tol) : st.zero_tol) * pmax) flag = shift_code;   // utils.c:354-356
        if (pp && flag > 0) {
            nprox = n;
            for (int i = lane; i < n; i += 64) b.prox_mask[(size_t)q * n + i] = 1;
        }
    }
    GPROF(1);
    // --- R -> R^-1, row by row as utils.c:380-389.  Lane <-> COLUMN: row k of the inverse lives in registers (one entry
    // per 64-column block and lane); step i takes t = x_i / r_ii from the lane that holds it and subtracts t times row i
    // of the Cholesky factor -- one coalesced load per block, shared by the KR rows of R^-1 that sweep together (their
    // chains are independent: each hides the other's broadcast latency).  Every entry still receives its terms in
    // ascending i, as in the reference.
    if (flag > 0) {
      if (!diag && !factored && doR) {
        constexpr int KR = 32 / NBK;
        for (int k0 = 0; k0 < n; k0 += KR) {
            double x[KR][NBK];
#pragma unroll
            for (int r = 0; r < KR; ++r) {      // utils.c:382-384: the diagonal 1/r_kk stays, the rest of row k times -(1/r_kk)
                const int k = k0 + r;
                const int pk = roff(k < n ? k : 0, n);
                const double rkk = (k < n) ? R[pk + k] : 0.0;
#pragma unroll
                for (int jb = 0; jb < NBK; ++jb) {
                    const int j = jb * 64 + lane;
                    const double v = (k < n && j < n && j >= k) ? R[pk + j] : 0.0;
                    x[r][jb] = (j > k) ? v * -rkk : v;
                }
            }
            for (int i0 = k0 + 1; i0 < n; i0 += 4) {   // four rows of the factor in flight
                double ri[4][NBK], rii[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int i = (i0 + u < n) ? i0 + u : n - 1;
                    const int pi = roff(i, n), bi = i >> 6;
                    rii[u] = R[pi + i];
#pragma unroll
                    for (int jb = 0; jb < NBK; ++jb) {
                        const int j = jb * 64 + lane;
                        ri[u][jb] = (jb >= bi && j > i && j < n) ? R[pi + j] : 0.0;
                    }
                }
                // (The column block of i picked by a wave-uniform branch around code with compile-time block numbers: blocks
                //  left of it are skipped, blocks right of it take the update without a select -- all their columns are > i --
                //  and x_i comes out of a register that is never indexed at run time.  The per-row "has this row started" test
                //  exists only in the first steps of a group and in the last group.)
                static_for<4>([&](auto uu) __attribute__((always_inline)) {
                    constexpr int u = uu;
                    const int i = i0 + u;
                    if (i < n) {
                        const int bi = i >> 6, li = i & 63;
                        const bool all_rows = (i >= k0 + KR) && (k0 + KR <= n);
                        static_for<NBK>([&](auto bb) __attribute__((always_inline)) {
                            if (bi == bb) {
                                auto row_step = [&](auto rr) __attribute__((always_inline)) {
                                    constexpr int r = rr;
                                    const double t = rl(x[r][bb], li) * rii[u];          // utils.c:386
                                    static_for<NBK>([&](auto jj) __attribute__((always_inline)) {
                                        constexpr int jb = jj;
                                        const int j = jb * 64 + lane;
                                        if constexpr (jb == bb) x[r][jb] = (j == i) ? t : ((j > i) ? x[r][jb] - ri[u][jb] * t : x[r][jb]);   // utils.c:387-388
                                        else if constexpr (jb > bb) x[r][jb] = x[r][jb] - ri[u][jb] * t;
                                    });
                                };
                                if (all_rows) static_for<KR>([&](auto rr) __attribute__((always_inline)) { row_step(rr); });
                                else static_for<KR>([&](auto rr) __attribute__((always_inline)) {
                                    const int k = k0 + rr;
                                    if (k < i && k < n) row_step(rr);                   // (rows of the group that have not started yet sit this step out)
                                });
                            }
                        });
                    }
                });
            }
#pragma unroll
            for (int r = 0; r < KR; ++r) {
                const int k = k0 + r;
                if (k < n) {
                    const int pk = roff(k, n);
#pragma unroll
                    for (int jb = 0; jb < NBK; ++jb) {
                        const int j = jb * 64 + lane;
                        if (j >= k && j < n) { Ro[pk + j] = x[r][jb]; if (mfma_m) Rsq[(size_t)k * sq_ld + j] = x[r][jb]; }
                    }
                }
            }
        }
        WSYNC();
      }
        GPROF(2);
        // --- v = R^-T f (utils.c:474-497, mask has UPDATE_Rinv: no column scaling)
        if (factored || !doV) {      // k_fact_wg formed it (and x_unc) from the R^-1 it had in LDS; or: neither R^-1 nor f changed
            for (int i = lane; i < n; i += 64) vv[i] = b.v[(size_t)q * n + i];
        } else
        for (int ic = 0; ic < n; ic += 64) {
            const int i = ic + lane;
            if (i < n) {
                double acc = Ro[roff(i, n) + i] * fl[i];
                for (int j = i - 1; j >= 0; --j) acc += Ro[roff(j, n) + i] * fl[j];
                vv[i] = acc;
            }
        }
        WSYNC();
    }
    // --- unconstrained optimum x = -R^-1 v (utils.c:618-662), only for the quadprog variant
    int unc = 0;
    if (flag > 0 && (mask & DAQP_UPDATE_unconstrained) && !(keepR && nprox > 0)) {   // (utils.c:622: not for a proximal workspace)
        int fixed = 0;
        for (int i = lane; i < m; i += 64) fixed |= sens[i] & (DAQP_ACTIVE + DAQP_IMMUTABLE);
        if (!__any(fixed)) {
            unc = 1;
            if (factored) {
                for (int i = lane; i < n; i += 64) xu[i] = b.xunc[(size_t)q * n + i];
            } else
            for (int ic = 0; ic < n; ic += 64) {
                const int i = ic + lane;
                if (i < n) {
                    const int pi = roff(i, n);
                    double s = 0;
                    for (int j = i; j < n; ++j) s += Ro[pi + j] * vv[j];
                    xu[i] = -s;
                }
            }
            WSYNC();
        }
    }
    // equality elimination would change the arithmetic of daqp_quadprog: not built (eq_elim.c:127-164)
    if (flag > 0 && (mask & DAQP_UPDATE_eliminate)) {
        int neq = 0;
        for (int base = ms; base < m; base += 64) {
            const int i = base + lane;
            const int isq = i < m && ((sens[i] & (DAQP_ACTIVE + DAQP_IMMUTABLE + DAQP_SOFT + DAQP_BINARY)) == (DAQP_ACTIVE + DAQP_IMMUTABLE));
            neq += __popcll(__ballot(isq));
        }
        if (neq > 5 && 10 * neq > n) flag = DAQP_EXIT_UNSUPPORTED;
    }

    // --- general rows: M = A R^-1 (utils.c:434-472), normalise (utils.c:586-613), d (utils.c:499-544
    // or, after the shortcut, utils.c:664-676 + 151-159).
    // Phase A forms the unnormalised rows, KB at a time, with lane <-> COLUMN: entry (k, c) is the reference's chain
    // R^-1[c][c] a[c] + R^-1[c-1][c] a[c-1] + ... + R^-1[0][c] a[0] in that order, so step r of all 64 chains of a column
    // block reads one contiguous stretch of row r of the packed R^-1 (a coalesced load, shared by the KB rows) and the KB
    // values a_k[r] as LDS broadcasts -- wherever the factors live (LDS or HBM scratch) nothing is fetched per lane and
    // per term any more.  Phase B (lane <-> row, the blocked image read back coalesced) normalises and forms d.
    int feasible = 1;
    double *Mq = b.Mblk + (size_t)q * b.nblk * b.npair * 128;
    // Default arithmetic mode: phase A on the matrix cores (v_mfma_f64_16x16x4_f64), 16 rows of A at a time.  The A operand of
    // a row tile (lane l: A[l&15][4kt + (l>>4)], every k step) is loaded from HBM once and stays in registers; the fragments of
    // R^-1 (lane l: R^-1[4kt + (l>>4)][16ct + (l&15)], zero below the diagonal) stream past it eight k steps ahead of the
    // matrix instructions that consume them, column tile by column tile, only up to each tile's last row (R^-1 is upper
    // triangular).  Nothing of A is staged in LDS.  The sums are fp64 fused in a different order than the reference's: M agrees
    // to ~1e-16 relative (the exact mode keeps the chain below).
    const bool defer_m = DEFER && mfma_m && b.defer_m != 0 && mA > 0;
    if (!DEFER && flag > 0 && mfma_m && !defer_m) {
        typedef double v4d __attribute__((ext_vector_type(4)));
        constexpr int KB = 16, NKT = 56;          // n <= 208: 52 k steps (56: whole blocks of eight)
        double *ob = smem + o.tile;               // [16][64] one column block of results on its way to the blocked image
        double2 *Mq2 = reinterpret_cast<double2 *>(Mq);
        const int lr = lane & 15, lk = lane >> 4;
        long long mp[3] = {0, 0, 0}, mt0 = b.prof ? (long long)__builtin_readcyclecounter() : 0;
#define MPROF(slot) do { if (b.prof) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); const long long t1 = (long long)__builtin_readcyclecounter(); mp[slot] += t1 - mt0; mt0 = t1; } } while (0)
        for (int kb = 0; kb < mA; kb += KB) {
            const int rows = (mA - kb) < KB ? (mA - kb) : KB;
            const bool rowok = lr < rows;
            const double *arow = A + (size_t)(kb + (rowok ? lr : 0)) * n;
            double av[NKT];
            // (straight-line per number of 32-column blocks of A, as for R^-1 below: all loads in flight, then the masks)
            static_for<NKT / 8>([&](auto nb) __attribute__((always_inline)) {
                if ((n + 31) / 32 == nb + 1) {
                    static_for<8 * (nb + 1)>([&](auto kt) __attribute__((always_inline)) {
                        const int kk = 4 * kt + lk;
                        av[kt] = arow[kk < n ? kk : 0];
                    });
                    static_for<8 * (nb + 1)>([&](auto kt) __attribute__((always_inline)) {
                        const int kk = 4 * kt + lk;
                        av[kt] = (rowok && kk < n) ? av[kt] : 0.0;
                    });
                    static_for<NKT - 8 * (nb + 1)>([&](auto kt) __attribute__((always_inline)) { av[8 * (nb + 1) + kt] = 0.0; });
                }
            });
            MPROF(0);
            for (int cb = 0; cb < n; cb += 64) {
                const int rtop = (n - 1 < cb + 63) ? n - 1 : cb + 63;
                WSYNC();   // the previous block's readers are done with ob
                for (int ct = 0; ct < 4 && cb + 16 * ct < n; ++ct) {
                    const int col = cb + 16 * ct + lr;
                    const bool colok = col < n;
                    const int klast = (n - 1 < cb + 16 * ct + 15) ? n - 1 : cb + 16 * ct + 15;
                    // (rows up to round_up(n,32) and columns up to round_up(n,16) of the square image exist and are zero outside
                    //  the upper triangle: no masks, no per-load address arithmetic beyond base + k step.  One straight-line
                    //  sequence per number of 32-row blocks this tile needs, chosen by a wave-uniform branch: with no control
                    //  flow between the loads and the matrix instructions the waits count exactly, all loads of the tile
                    //  are in flight before the first matrix instruction and each waits for its own operand only.)
                    const double *bcol = Rsq + (size_t)lk * sq_ld + col;
                    const int nblk = klast / 32 + 1;
                    v4d acc = (v4d){0.0, 0.0, 0.0, 0.0};
                    static_for<NKT / 8>([&](auto nb) __attribute__((always_inline)) {
                        if (nblk == nb + 1) {
                            double bq[nb + 1][8];
                            static_for<nb + 1>([&](auto c8) __attribute__((always_inline)) {
                                static_for<8>([&](auto u) __attribute__((always_inline)) { bq[c8][u] = bcol[(size_t)(32 * c8 + 4 * u) * sq_ld]; });
                            });
                            static_for<nb + 1>([&](auto c8) __attribute__((always_inline)) {
                                static_for<8>([&](auto u) __attribute__((always_inline)) {
                                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[8 * c8 + u], bq[c8][u], acc, 0, 0, 0);
                                });
                            });
                        }
                    });
                    static_for<4>([&](auto r) __attribute__((always_inline)) { ob[(lk + 4 * r) * 64 + 16 * ct + lr] = acc[(int)r]; });   // D: row (l>>4) + 4 reg, column l&15
                }
                WSYNC();
                MPROF(1);
                const int pairs = ((rtop - cb) >> 1) + 1;
                for (int idx = lane; idx < pairs * KB; idx += 64) {
                    const int k = idx & (KB - 1), t = idx / KB;
                    if (k < rows) {
                        const int gi = ms + kb + k;
                        double2 vpair;
                        vpair.x = ob[k * 64 + 2 * t];
                        vpair.y = (cb + 2 * t + 1 < n) ? ob[k * 64 + 2 * t + 1] : 0.0;
                        Mq2[((size_t)(gi >> 6) * b.npair + (cb >> 1) + t) * 64 + (gi & 63)] = vpair;
                    }
                }
                MPROF(2);
            }
        }
        if (b.prof && lane == 0) for (int i = 0; i < 3; ++i) b.prof[(size_t)q * 32 + 22 + i] = mp[i];
#undef MPROF
    }
    if ((!DEFER || !defer_m) && flag > 0 && !defer_m) {
        constexpr int KB = kSetupRows;
        const int np2 = round_up(n, 2);
        double *at = smem + o.tile;               // [KB][np2] rows of A
        double *ob = at + KB * np2;               // [KB][64] one column block of results on its way to the blocked image
        double2 *Mq2 = reinterpret_cast<double2 *>(Mq);
        for (int kb = 0; kb < mA && !mfma_m; kb += KB) {
            const int rows = (mA - kb) < KB ? (mA - kb) : KB;
            WSYNC();
            stage_rows(at, A + (size_t)kb * n, rows, n, np2);
            for (int e = lane; e < (KB - rows) * np2; e += 64) at[rows * np2 + e] = 0.0;
            WSYNC();
            if (unc && lane < rows) {             // A_k . x_unc, parked in dupper until phase B (utils.c:664-668)
                double sunc = 0;
                for (int j = 0; j < n; ++j) sunc += at[lane * np2 + j] * xu[j];
                du[ms + kb + lane] = sunc;
            }
            if constexpr (PART) if (keepR && !diag && ms > 0) {   // utils.c:447-452: the kept R^-1 has its rows < ms normalised
                WSYNC();
                for (int e = lane; e < rows * ms; e += 64) { const int k = e / ms, c = e - k * ms; at[k * np2 + c] = at[k * np2 + c] / sc[c]; }
                WSYNC();
            }
            for (int cb = 0; cb < n; cb += 64) {
                const int c = cb + lane;
                const bool has = c < n;
                // A chain starts with an assignment (utils.c:444).  -0.0 + p == p bit for bit for every p (signed zeros
                // included), so the accumulators start at -0.0 and the first term is added like the others.
                double acc[KB];
#pragma unroll
                for (int k = 0; k < KB; ++k) acc[k] = -0.0;
                const int rtop = (n - 1 < cb + 63) ? n - 1 : cb + 63;
                // the diagonal block: a lane joins its chain at r == c, lanes right of it accumulate, lanes left of it sit out
                for (int r0 = rtop; r0 >= cb; r0 -= 8) {
                    double rv[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) { const int r = r0 - u; rv[u] = (has && r >= cb && r <= c) ? Ro[roff(r, n) + c] : 0.0; }
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int r = r0 - u;
                        if (r >= cb) {
                            if (has && r <= c) {
#pragma unroll
                                for (int k = 0; k < KB; ++k) acc[k] += rv[u] * at[k * np2 + r];
                            }
                        }
                    }
                }
                // above it every lane of the block takes part: 16 rows of R^-1 in flight, then KB x 16 multiply-adds
                for (int rb = (cb >> 4) - 1; rb >= 0; --rb) {   // rows 16 rb + 15 ... 16 rb (cb is a multiple of 64)
                    double rv[16];
#pragma unroll
                    for (int u = 0; u < 16; ++u) rv[u] = has ? Ro[roff(16 * rb + 15 - u, n) + c] : 0.0;
#pragma unroll
                    for (int k = 0; k < KB; ++k) {
                        const double2 *ap = reinterpret_cast<const double2 *>(at + k * np2 + 16 * rb);   // 128-byte aligned: eight 16-byte broadcasts
#pragma unroll
                        for (int h = 7; h >= 0; --h) {
                            const double2 a2 = ap[h];
                            acc[k] += rv[15 - (2 * h + 1)] * a2.y;
                            acc[k] += rv[15 - 2 * h] * a2.x;
                        }
                    }
                }
                // through LDS into the blocked image [row/64][col/2][row%64][col%2]: 16 rows x 16 bytes contiguous per column pair
                WSYNC();
#pragma unroll
                for (int k = 0; k < KB; ++k) ob[k * 64 + lane] = has ? acc[k] : 0.0;
                WSYNC();
                const int pairs = ((rtop - cb) >> 1) + 1;
                for (int idx = lane; idx < pairs * KB; idx += 64) {
                    const int k = idx & (KB - 1), t = idx / KB;
                    if (k < rows) {
                        const int gi = ms + kb + k;
                        double2 vpair;
                        vpair.x = ob[k * 64 + 2 * t];
                        vpair.y = (cb + 2 * t + 1 < n) ? ob[k * 64 + 2 * t + 1] : 0.0;
                        Mq2[((size_t)(gi >> 6) * b.npair + (cb >> 1) + t) * 64 + (gi & 63)] = vpair;
                    }
                }
            }
        }
        // phase B reads the image back through other lanes of this ONE wave: a workgroup-scope fence (a device-scope fence writes the
        // XCD's dirty L2 back -- per problem; setup_m.hip.h has the measurement)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        GPROF(3);
        WSYNC();
        for (int tb = 0; tb < mA && flag > 0; tb += 64) {
            const int k = tb + lane;
            const bool own = k < mA;
            const int gi = ms + (own ? k : 0);
            double2 *rowp = Mq2 + ((size_t)(gi >> 6) * b.npair) * 64 + (gi & 63);
            // fp32 image [row/64][col/4][row%64][col%4] for the workgroup kernel's screening scan, as float2 halves
            float2 *row32 = b.M32 ? reinterpret_cast<float2 *>(b.M32 + (size_t)q * b.nblk * b.nquad * 256) + (((size_t)(gi >> 6) * b.nquad) * 64 + (gi & 63)) * 2 : nullptr;
            int rowbad = 0;
            if (own) {
                double s = 0;
                for (int t0 = 0; t0 < b.npair; t0 += 8) {   // ||M_k||^2 in column order, 8 loads in flight
                    double2 v8[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) v8[u] = rowp[(size_t)((t0 + u < b.npair) ? t0 + u : 0) * 64];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int t = t0 + u;
                        if (t < b.npair) { s += v8[u].x * v8[u].x; if (2 * t + 1 < n) s += v8[u].y * v8[u].y; }
                    }
                }
                double scal = 1.0;
                bool scale_it = true;
                if (s < st.zero_tol) {
                    scale_it = false;
                    if (bu[gi] < -st.zero_tol || bl[gi] > st.zero_tol)
                        if (!(sens[gi] & DAQP_IMMUTABLE) && !(sens[gi] & DAQP_SOFT)) rowbad = 1;
                    sens[gi] = DAQP_IMMUTABLE;
                } else scal = 1 / sqrt(s);
                double dsum = 0, sraw = 0;
                for (int t0 = 0; t0 < b.npair; t0 += 8) {
                    double2 v8[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) v8[u] = rowp[(size_t)((t0 + u < b.npair) ? t0 + u : 0) * 64];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int t = t0 + u;
                        if (t < b.npair) {
                            double2 w = v8[u];
                            if (unc && mfma_m) { sraw += w.x * vv[2 * t]; if (2 * t + 1 < n) sraw += w.y * vv[2 * t + 1]; }
                            if (scale_it) { w.x *= scal; if (2 * t + 1 < n) w.y *= scal; }
                            if (!unc) { dsum += w.x * vv[2 * t]; if (2 * t + 1 < n) dsum += w.y * vv[2 * t + 1]; }
                            if (scale_it) rowp[(size_t)t * 64] = w;
                            if (row32) row32[((size_t)(t >> 1) * 64) * 2 + (t & 1)] = make_float2((float)w.x, (2 * t + 1 < n) ? (float)w.y : 0.0f);
                        }
                    }
                }
                sc[gi] = scal;
                if (unc) {
                    // A_k . x_unc: parked by the exact phase A; the matrix-core phase has no row of A in LDS and uses
                    // A x_unc = -(A R^-1) v, the unnormalised row against v (equal up to rounding)
                    const double sunc = mfma_m ? -sraw : du[gi];
                    const double u0 = bu[gi] - sunc, l0 = bl[gi] - sunc;
                    if (u0 < -st.primal_tol || l0 > st.primal_tol) feasible = 0;
                    du[gi] = u0 * scal; dl[gi] = l0 * scal;
                } else {
                    du[gi] = bu[gi] * scal + dsum;
                    dl[gi] = bl[gi] * scal + dsum;
                }
            }
            if (__any(rowbad)) flag = DAQP_EXIT_INFEASIBLE; // reference returns at the first such row
        }
    }
    GPROF(4);
    // --- simple bounds: normalise rows < ms of R^-1 (utils.c:569-585), their d, and their dense image in M
    if (flag > 0) {
        WSYNC();
        for (int ic = 0; ic < ms; ic += 64) {
            const int i = ic + lane;
            if (i < ms) {
                const int pi = roff(i, n);
                double s = 0;
                if (diag || keepR) s = sc[i];   // sqrt(H_ii): the row stays un-normalised and counts as the unit vector; or the kept normalisation
                else {
                    for (int j = i; j < n; ++j) s += Ro[pi + j] * Ro[pi + j];
                    s = 1 / sqrt(s);
                    sc[i] = s;
                    for (int j = i; j < n; ++j) Ro[pi + j] *= s;
                }
                if (unc) {
                    const double u0 = bu[i] - xu[i], l0 = bl[i] - xu[i];
                    if (u0 < -st.primal_tol || l0 > st.primal_tol) feasible = 0;
                    du[i] = u0 * s; dl[i] = l0 * s;
                } else {
                    double t = 0;
                    if (diag) t = vv[i];
                    else for (int j = i; j < n; ++j) t += Ro[pi + j] * vv[j];
                    du[i] = bu[i] * s + t;
                    dl[i] = bl[i] * s + t;
                }
                double2 *dst = reinterpret_cast<double2 *>(Mq) + ((size_t)(i >> 6) * b.npair) * 64 + (i & 63);
                if (!keepR)
                for (int t = 0; t < b.npair; ++t) {
                    double2 vpair;
                    if (diag) { vpair.x = (2 * t == i) ? 1.0 : 0.0; vpair.y = (2 * t + 1 == i) ? 1.0 : 0.0; }
                    else {
                        vpair.x = (2 * t >= i) ? Ro[pi + 2 * t] : 0.0;
                        vpair.y = (2 * t + 1 >= i && 2 * t + 1 < n) ? Ro[pi + 2 * t + 1] : 0.0;
                    }
                    dst[(size_t)t * 64] = vpair;
                    if (b.M32) {
                        float2 *d32 = reinterpret_cast<float2 *>(b.M32 + (size_t)q * b.nblk * b.nquad * 256) + (((size_t)(i >> 6) * b.nquad) * 64 + (i & 63)) * 2;
                        d32[((size_t)(t >> 1) * 64) * 2 + (t & 1)] = make_float2((float)vpair.x, (float)vpair.y);
                    }
                }
            }
        }
        WSYNC();
    }
    const int all_feasible = __all(feasible);
    int sing = kEmpty;
    if (flag > 0 && unc && all_feasible && !defer_m) { sing = DAQP_UNCONSTRAINED_OPTIMAL; activate = 0; }
    // general rows deferred to k_setup_m: what it has to know (the shortcut is decided there, once every row's d is known)
    const int owed = (flag > 0 && defer_m) ? (1 | (unc ? 2 : 0) | (all_feasible ? 0 : 4)) : 0;
    // --- write back
    if (flag > 0) {
        if (!keepR) for (int e = lane; e < b.rtri; e += 64) b.Rinv[(size_t)q * b.rtri + e] = Ro[e];
        for (int i = lane; i < n; i += 64) { b.v[(size_t)q * n + i] = vv[i]; if (unc) b.xunc[(size_t)q * n + i] = xu[i]; }
    }
    for (int i = lane; i < m; i += 64) b.sense[(size_t)q * m + i] = sens[i];
    if constexpr (PART) {
        // what fails the reference's update without touching the factor (a zero row whose bounds exclude 0, utils.c:600-602) is the
        // update's flag (k_update): the workspace stays set up and a later update repairs it.  A Hessian that cannot be factorised
        // leaves no workspace to solve on: that is the setup flag, as after a setup.
        const bool hard = flag == DAQP_NEEDS_SHIFT || flag == DAQP_EXIT_NONCONVEX || flag == DAQP_EXIT_UNSUPPORTED;
        if (lane == 0) {
            qs->n_active = 0; qs->reuse_ind = 0; qs->sing_ind = sing; qs->lam_swapped = 0; qs->pad_ = 0;
            if (doR) {
                const int sf = (flag > 0 || hard) ? flag : 1;
                qs->setup_flag = sf; qs->exitflag = sf; qs->iterations = 0; qs->fval = 0; qs->soft_slack = 0;
                qs->diag_h = diag; qs->n_prox = (flag > 0) ? nprox : 0;
            }
            qs->upd_flag = (flag < 0 && !hard) ? flag : 0;
            qs->need_activate = (flag > 0 || flag == DAQP_NEEDS_SHIFT) ? activate : 0;   // (the regularising re-run picks it up)
        }
    } else
    if (lane == 0) {
        qs->n_active = 0; qs->reuse_ind = 0; qs->sing_ind = sing; qs->iterations = 0;
        qs->lam_swapped = 0; qs->setup_flag = flag; qs->need_activate = (flag > 0) ? activate : 0;
        qs->exitflag = flag; qs->fval = 0; qs->soft_slack = 0; qs->diag_h = diag; qs->n_prox = (flag > 0) ? nprox : 0;
        qs->upd_flag = 0; qs->pad_ = owed;
    }
    GPROF(5);
    if (b.prof && lane == 0) for (int i = 0; i < 6; ++i) b.prof[(size_t)q * 32 + 16 + i] = gpt[i];
#undef GPROF
}

// ------------------------------------------------------------------------------------
// k_mirror: everything a single-problem workspace shows its caller on the host (types.h:187-264: the iterate's scalars,
// WS, sense, lam_star and, with `ldp`, v, d, scaling, R^-1 and M in the reference's row-major layout) gathered into ONE
// slab, so that the host takes it in one copy instead of ten.  Problem 0 of the batch.
//   doubles: [0,16) QState bytes | lam_star[cap] | v[n] dupper[m] dlower[m] scaling[m] Rinv[rtri] M[(m-ms) n] | ints: WS[cap] sense[m]
// ------------------------------------------------------------------------------------
__host__ __device__ inline size_t mirror_ldp_off(int cap) { return 16 + (size_t)cap; }
__host__ __device__ inline size_t mirror_int_off(int n, int m, int ms, int cap, int rtri) { return mirror_ldp_off(cap) + n + 3 * (size_t)m + rtri + (size_t)(m - ms) * n; }
__host__ __device__ inline size_t mirror_doubles(int n, int m, int ms, int cap, int rtri) { return mirror_int_off(n, m, ms, cap, rtri) + (size_t)(cap + m + 1) / 2 + 1; }
__global__ __launch_bounds__(256) void k_mirror(BatchDev b, double *slab, int ldp)
{
    const int T = gridDim.x * blockDim.x, t0 = blockIdx.x * blockDim.x + threadIdx.x;
    const int n = b.n, m = b.m, ms = b.ms, cap = b.cap;
    const QState *qs = b.qs;
    if (t0 < (int)(sizeof(QState) / sizeof(int))) reinterpret_cast<int *>(slab)[t0] = reinterpret_cast<const int *>(qs)[t0];
    const double *lam = b.vecs + (qs->lam_swapped ? 3 : 4) * (size_t)cap;
    for (int i = t0; i < cap; i += T) slab[16 + i] = lam[i];
    int *io = reinterpret_cast<int *>(slab + mirror_int_off(n, m, ms, cap, b.rtri));
    for (int i = t0; i < cap; i += T) io[i] = b.WS[i];
    for (int i = t0; i < m; i += T) io[cap + i] = b.sense[i];
    if (!ldp) return;
    double *o = slab + mirror_ldp_off(cap);
    for (int i = t0; i < n; i += T) o[i] = b.v[i];
    o += n;
    for (int i = t0; i < m; i += T) { o[i] = b.dupper[i]; o[m + i] = b.dlower[i]; o[2 * m + i] = b.scaling[i]; }
    o += 3 * (size_t)m;
    if (ldp < 2) return;
    for (int i = t0; i < b.rtri; i += T) o[i] = b.Rinv[i];
    o += b.rtri;
    const int mA = m - ms;
    for (size_t e = t0; e < (size_t)mA * n; e += T) {
        const int r = ms + (int)(e / n), k = (int)(e % n);
        o[e] = b.Mblk[(((size_t)(r >> 6) * b.npair + (k >> 1)) * 64 + (r & 63)) * 2 + (k & 1)];
    }
}

// ------------------------------------------------------------------------------------
// k_update: DAQP_UPDATE_v and/or DAQP_UPDATE_d on an existing LDP (utils.c:58-221 with those masks)
// ------------------------------------------------------------------------------------
// daqp_batch_setup_shared: per-problem state after the ONE factorisation (done with wide-open bounds, so the sense it
// left holds only the structural bits: rows of A R^-1 that vanish, utils.c:586-613).  One wave per problem.
__global__ __launch_bounds__(64) void k_init_shared(BatchDev b, const int *structural, const int *shared_flag /* [setup_flag, diag_h, n_prox] */)
{
    const int q = blockIdx.x, lane = lane_id(), m = b.m;
    const double *bu = b.bu + (size_t)q * m, *bl = b.bl + (size_t)q * m;
    int flag = *shared_flag, bad = 0;
    for (int i = lane; i < m; i += 64) {
        int s = b.sense_in ? b.sense_in[(size_t)q * m + i] : 0;
        if (s & DAQP_BINARY) bad |= 2;
        if (structural[i] & DAQP_IMMUTABLE) {   // zero row: feasible only if 0 lies within its bounds (utils.c:598-606)
            if (bu[i] < -b.st.zero_tol || bl[i] > b.st.zero_tol)
                if (!(s & DAQP_IMMUTABLE) && !(s & DAQP_SOFT)) bad |= 1;
            s = DAQP_IMMUTABLE;
        }
        b.sense[(size_t)q * m + i] = s;
    }
    if (flag > 0 && __any(bad & 2)) flag = DAQP_EXIT_UNSUPPORTED;
    else if (flag > 0 && __any(bad & 1)) flag = DAQP_EXIT_INFEASIBLE;
    if (lane == 0) {
        QState *qs = b.qs + q;
        qs->n_active = 0; qs->reuse_ind = 0; qs->sing_ind = kEmpty; qs->iterations = 0;
        qs->lam_swapped = 0; qs->setup_flag = flag; qs->need_activate = (flag > 0 && b.sense_in) ? 1 : 0; qs->pad_ = 0;
        qs->exitflag = flag; qs->fval = 0; qs->soft_slack = 0; qs->diag_h = shared_flag[1]; qs->n_prox = shared_flag[2];
        qs->upd_flag = 0;
    }
}

__global__ __launch_bounds__(64) void k_update(BatchDev b, int mask)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int q = blockIdx.x, lane = lane_id();
    const int n = b.n, m = b.m, ms = b.ms;
    double *vv = smem, *fl = smem + round_up(n, 2);
    QState *qs = b.qs + q;
    if (qs->setup_flag < 0) return;
    const double *bu = b.bu + (size_t)q * m, *bl = b.bl + (size_t)q * m;
    const double *Rq = b.Rinv + qf(b, q) * b.rtri;
    const double *scq = b.scaling + qf(b, q) * m;
    int *sens = b.sense + (size_t)q * m;
    const DAQPSettings &st = b.st;
    // check_bounds (utils.c:546-567) on the stored sense.  The reference walks the rows in order and returns at the first
    // crossed pair: unmarked equalities BEFORE it have been marked by then, nothing else of the workspace has changed
    // (utils.c:98-103 comes before v, d and the activation) -- the workspace stays usable and the next update starts afresh.
    int first_bad = kBig;
    for (int i = lane; i < m; i += 64)
        if (!(sens[i] & DAQP_IMMUTABLE) && bu[i] - bl[i] < -st.primal_tol && first_bad == kBig) first_bad = i;
    first_bad = (int)wave_min((double)first_bad);   // lowest index over the wave (exact in fp64)
    int bad = 0;
    for (int i = lane; i < m && i < first_bad; i += 64) {
        const int s = sens[i];
        if (!(s & DAQP_IMMUTABLE) && !(s & DAQP_SOFT) && bu[i] - bl[i] < st.zero_tol) { sens[i] = s | DAQP_ACTIVE | DAQP_IMMUTABLE; bad |= 4; }
    }
    const int inf = first_bad != kBig, eq = __any(bad & 4);
    if (lane == 0) {
        qs->upd_flag = inf ? DAQP_EXIT_INFEASIBLE : 0; qs->sing_ind = kEmpty;   // utils.c:80-81 precedes the check
        if (inf) qs->need_activate = 0;   // (a sense taken over by the same call asked for the activation: the reference returns before it, utils.c:95-96)
    }
    if (inf) return;
    if (mask & DAQP_UPDATE_v) {   // utils.c:474-497 without UPDATE_Rinv: rows < ms of R^-1 are normalised
        const double *f = b.f + (size_t)q * n;
        const int diag = qs->diag_h;   // RinvD branch: v_i = f_i * RinvD_i, no scaling (utils.c:479-480)
        for (int i = lane; i < n; i += 64) fl[i] = (i < ms && !diag) ? f[i] / scq[i] : f[i];
        WSYNC();
        for (int ic = 0; ic < n; ic += 64) {
            const int i = ic + lane;
            if (i < n) {
                double acc = Rq[roff(i, n) + i] * fl[i];
                for (int j = i - 1; j >= 0; --j) acc += Rq[roff(j, n) + i] * fl[j];
                vv[i] = acc;
                b.v[(size_t)q * n + i] = acc;
            }
        }
    } else {
        for (int i = lane; i < n; i += 64) vv[i] = b.v[(size_t)q * n + i];
    }
    WSYNC();
    // d = b*scaling + (row . v) for all m rows of the dense blocked image (utils.c:499-544)
    const double2 *v2 = reinterpret_cast<const double2 *>(vv);
    const bool odd = (n & 1) != 0;
    const int full = odd ? b.npair - 1 : b.npair;
    const double *Mq = b.Mblk + qf(b, q) * b.nblk * b.npair * 128;
    for (int blk = 0; blk < b.nblk; ++blk) {
        const int r = blk * 64 + lane;
        if (r < m) {
            const double2 *src = reinterpret_cast<const double2 *>(Mq) + ((size_t)blk * b.npair) * 64 + lane;
            double s = 0;
            for (int t = 0; t < full; ++t) {
                const double2 mm = src[(size_t)t * 64];
                const double2 vk = v2[t];
                s += mm.x * vk.x;
                s += mm.y * vk.y;
            }
            if (odd) s += src[(size_t)full * 64].x * vv[n - 1];
            b.dupper[(size_t)q * m + r] = bu[r] * scq[r] + s;
            b.dlower[(size_t)q * m + r] = bl[r] * scq[r] + s;
        }
    }
    if (lane == 0) {
        qs->reuse_ind = 0;
        qs->sing_ind = kEmpty;   // utils.c:80-81
        if (eq) qs->need_activate = 1;
    }
}

// k_update_sense: the DAQP_UPDATE_sense step of daqp_update_ldp on its own (utils.c:80-91): the caller's sense replaces the
// workspace's -- bits that an earlier bound check or a vanishing row of A R^-1 had set are gone with it, as in the reference -- or
// zeros when the caller has none; a given sense asks for the working set to be rebuilt from its ACTIVE bits (utils.c:199-211, the
// activation launch that follows).  One workgroup per problem.
__global__ __launch_bounds__(256) void k_update_sense(BatchDev b)
{
    const int q = blockIdx.x, m = b.m;
    QState *qs = b.qs + q;
    int binary = 0;
    if (b.sense_in) for (int i = threadIdx.x; i < m; i += 256) binary |= b.sense_in[(size_t)q * m + i] & DAQP_BINARY;
    if (__syncthreads_or(binary)) {            // binary constraints are outside this path: nothing is taken over
        if (threadIdx.x == 0) qs->upd_flag = DAQP_EXIT_UNSUPPORTED;
        return;
    }
    for (int i = threadIdx.x; i < m; i += 256) b.sense[(size_t)q * m + i] = b.sense_in ? b.sense_in[(size_t)q * m + i] : 0;
    if (threadIdx.x == 0) {
        qs->sing_ind = kEmpty;                 // utils.c:80-81
        qs->upd_flag = 0;
        if (b.sense_in) qs->need_activate = 1;
    }
}

// ------------------------------------------------------------------------------------
// k_ldp: mode 0 = daqp_solve, mode 1 = only (re)build the working set from the ACTIVE bits
// (the tail of daqp_update_ldp, utils.c:199-211)
// ------------------------------------------------------------------------------------
// LDS (L + active-row cache) already limits residency to one wave per SIMD, so the register
// variants may take the whole unified 512-entry VGPR/AGPR file: waves_per_eu(1, NB > 0 ? 1 : 8)
#ifndef DAQP_AMD_SPILL_WAVES
#define DAQP_AMD_SPILL_WAVES 2
#endif
// The spilled variant (L and the active-row cache in HBM, n = 200 class): 2 waves per SIMD measured best (3, 4 and 5 waves
// cost more in scratch spills than they hide in latency: 604 / 613 / 635 ms per 8 192 QPs of config C4).
template <int C, bool SPILL, int NB, int NP>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(SPILL ? DAQP_AMD_SPILL_WAVES : 1, SPILL ? DAQP_AMD_SPILL_WAVES : 8)))
void k_ldp(BatchDev b, int mode_in)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int q = blockIdx.x, lane = lane_id();
    const int n = b.n, m = b.m, cap = b.cap;
    // mode | 4: behind the workgroup kernel (wg_kernel.hip.h) -- only the problems it flagged as too large for its LDS
    if ((mode_in & 4) && !b.fallback[q]) return;
    const int mode = mode_in & 3;
    QState *qs = b.qs + q;
    const int sflag = qs->setup_flag;
    if (mode == 1) { if (sflag < 0 || !qs->need_activate) return; }
    if (sflag < 0) {   // setup failed: x/lam untouched, no solve (api.c:70-78)
        if (lane == 0) { b.exitflag[q] = sflag; b.iter[q] = 0; if (b.fval) b.fval[q] = 0; if (b.soft) b.soft[q] = 0; }
        return;
    }
    if (mode == 0 && qs->upd_flag < 0) {   // the last update failed its bound check: report that, keep the state (see k_update)
        if (lane == 0) { b.exitflag[q] = qs->upd_flag; b.iter[q] = 0; if (b.fval) b.fval[q] = 0; if (b.soft) b.soft[q] = 0; }
        return;
    }
    const LdpLds o = ldp_lds(n, m, cap, SPILL);
    int *ibase = reinterpret_cast<int *>(smem + o.dbl);
    Wave<C, NB, NP> w;
    w.t_start = solve_stamp(b.tstart, q); w.tick_s = b.tick_s;
    w.profiling = (b.prof != nullptr) && mode == 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) w.prof[i] = 0;
    w.n = n; w.m = m; w.ms = b.ms; w.cap = cap; w.npair = b.npair; w.nblk = b.nblk; w.ldr = b.ldr;
    if (SPILL) {
        w.L = b.L + (size_t)q * b.ltri;
        w.rowc = b.rowc_g + (size_t)q * cap * b.ldr;
    } else {
        w.L = smem + o.L;
        w.rowc = smem + o.rowc;
    }
    w.D = smem + o.D; w.xl = smem + o.xl; w.zl = smem + o.zl;
    double *lamA = smem + o.lamA, *lamB = smem + o.lamB;
    w.u = smem + o.u; w.pend_lam = smem + o.pend_lam;
    w.ws = ibase + o.ws; w.sense = ibase + o.sense; w.pend_id = ibase + o.pend_id;
    w.Mblk = b.Mblk + qf(b, q) * b.nblk * b.npair * 128;
    w.dupper = b.dupper + (size_t)q * m; w.dlower = b.dlower + (size_t)q * m; w.scaling = b.scaling + qf(b, q) * m;
    w.st = b.st;
    w.trace = b.trace ? b.trace + (size_t)q * b.trace_cap : nullptr;
    w.trace_cap = b.trace_cap; w.trace_len = 0;
    w.na = qs->n_active; w.reuse = qs->reuse_ind; w.sing = qs->sing_ind;
    w.fval = qs->fval; w.soft = qs->soft_slack;
    const int swapped = qs->lam_swapped;
    w.lam = swapped ? lamB : lamA; w.lams = swapped ? lamA : lamB;

    // ---- load the persistent iterate
    int *gsense = b.sense + (size_t)q * m;
    int softbits = 0;
    for (int i = lane; i < m; i += 64) { const int s = gsense[i]; w.sense[i] = s; softbits |= s & DAQP_SOFT; }
    w.has_soft = __any(softbits) ? 1 : 0;
    double *gv = b.vecs + (size_t)q * 5 * cap;
    int *gws = b.WS + (size_t)q * cap;
    if (w.sing != DAQP_UNCONSTRAINED_OPTIMAL) {
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
        for (int i = 0; i < w.na; ++i) fetch_row(w, w.ws[i], i);   // rebuild the active-row cache
    }
    WSYNC();

    int flag = 1, iters = 0;
    if (mode == 1) {
        reset_ws(w);
        flag = activate_marked(w);
        if (lane == 0) { qs->need_activate = 0; if (flag < 0) { qs->setup_flag = flag; qs->exitflag = flag; } }
    } else if (w.sing == DAQP_UNCONSTRAINED_OPTIMAL) {
        // api.c:40-45: x = unconstrained optimum, no multipliers
        const double *xu = b.xunc + (size_t)q * n, *vq = b.v + (size_t)q * n;
        if (b.x) for (int i = lane; i < n; i += 64) b.x[(size_t)q * n + i] = xu[i];
        if (b.lam) for (int i = lane; i < m; i += 64) b.lam[(size_t)q * m + i] = 0;
        double fv = 0;
        for (int i = 0; i < n; ++i) { const double vi = vq[i]; fv -= vi * vi; }
        fv *= 0.5;
        if (lane == 0) {
            b.exitflag[q] = DAQP_EXIT_OPTIMAL; b.iter[q] = 1;
            if (b.fval) b.fval[q] = fv;
            if (b.soft) b.soft[q] = 0;
            qs->iterations = 1; qs->fval = 0; qs->soft_slack = 0; qs->exitflag = DAQP_EXIT_OPTIMAL;
        }
        return;
    } else {
        if (qs->need_activate) {   // defensive: setup/update normally runs mode 1 itself
            reset_ws(w);
            flag = activate_marked(w);
        }
        if (flag >= 0) flag = ldp_loop(w, iters);
        // ---- ldp2qp_solution (daqp.c:111-139) + daqp_extract_result (api.c:455-495)
        const double *Rq = b.Rinv + qf(b, q) * b.rtri, *vq = b.v + (size_t)q * n;
        if (flag > 0) {
            for (int i = lane; i < n; i += 64) w.u[i] = w.u[i] - vq[i];
            WSYNC();
            for (int ic = 0; ic < n; ic += 64) {
                const int i = ic + lane;
                if (i < n) {
                    const double *row = Rq + roff(i, n);
                    double xi = w.u[i] * row[i];
                    for (int j = i + 1; j < n; ++j) xi += row[j] * w.u[j];
                    if (i < b.ms && !qs->diag_h) xi /= w.scaling[i];   // daqp.c:124-134: no division in the RinvD branch
                    if (b.x) b.x[(size_t)q * n + i] = xi;
                }
            }
            for (int i = lane; i < w.na; i += 64) w.lams[i] *= w.scaling[w.ws[i]];
            WSYNC();
        } else if (b.x) {
            // the reference copies whatever work->x holds; give the LDP iterate back as is
            for (int i = lane; i < n; i += 64) b.x[(size_t)q * n + i] = w.u[i];
        }
        if (b.lam) {
            for (int i = lane; i < m; i += 64) b.lam[(size_t)q * m + i] = 0;
            WSYNC();
            for (int i = lane; i < w.na; i += 64) b.lam[(size_t)q * m + w.ws[i]] = w.lams[i];
        }
        double fv = w.fval;
        for (int i = 0; i < n; ++i) { const double vi = vq[i]; fv -= vi * vi; }
        fv *= 0.5;
        if (lane == 0) {
            b.exitflag[q] = flag; b.iter[q] = iters;
            if (b.fval) b.fval[q] = fv;
            if (b.soft) b.soft[q] = w.soft;
            qs->iterations = iters; qs->exitflag = flag; qs->need_activate = 0;
        }
    }
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
        qs->fval = w.fval; qs->soft_slack = w.soft;
        if (b.trace) b.trace[(size_t)q * b.trace_cap + b.trace_cap - 1] = w.trace_len;
        if (w.profiling)
            for (int i = 0; i < 8; ++i) b.prof[(size_t)q * 32 + i] = w.prof[i];
    }
}

} // namespace daqp_amd
