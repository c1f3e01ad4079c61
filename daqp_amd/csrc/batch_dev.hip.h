// batch_dev.hip.h -- the device-side descriptor of a batch (what every kernel of the library receives)
#pragma once
#include "wave_ldp.hip.h"

namespace daqp_amd {


struct BatchDev {
    int N, n, m, ms, cap, mA;
    int npair, nblk, ldr, ltri, rtri;
    int ldrc;   // row stride of the register kernel's active-row cache: == 2 (mod 4) doubles, >= 2*NP (16-B rows, conflict-free)
    // problem data (device pointers; owned by the caller or by the batch's staging buffers)
    const double *H, *f, *A, *bu, *bl;
    const int *sense_in;
    // LDP (persistent)
    double *Mblk;      // [N][nblk][npair][64][2]
    float *M32;        // [N][nblk][nquad][64][4] the same image rounded to fp32 (workgroup kernel's screening scan), or null
    int nquad;         // (npair + 1) / 2
    int wg_inverse;    // workgroup kernel, default arithmetic: carry L^-1 instead of L (wg_ldp.hip.h)
    double *Rinv;      // [N][rtri]   packed upper R^-1, rows < ms normalised
    double *v;         // [N][n]
    double *scaling, *dupper, *dlower; // [N][m]
    int *sense;        // [N][m]
    double *xunc;      // [N][n]   unconstrained optimum when the shortcut fired
    // iterate (persistent: warm start)
    double *L;         // [N][ltri]
    double *vecs;      // [N][5][cap]: D, xldl, zldl, lam buffer A, lam buffer B
    int *WS;           // [N][cap]
    QState *qs;        // [N]
    double *rowc_g;    // [N][cap*ldr] only when the active-row cache / L spill out of LDS
    double *setup_g;   // [N][2*rtri] only when the setup factors spill out of LDS
    double *setup_sq;  // [N][round_up(n,32)][round_up(n,16)] generic setup, default mode: R^-1 as a zero-padded square (the matrix cores' B operand)
    int defer_m;       // 1: k_setup leaves the general rows (M = A R^-1, normalisation, d, images) to k_setup_m, launched right behind it
    double *fact;      // [N][4] k_fact_wg (setup_fact.hip.h) ran in front of k_setup: {1 = R^-1 is in setup_g / setup_sq already, smallest pivot, largest pivot}, or null
    int *m_tick;       // [N] k_setup_m: per problem, workgroups done and how many of them saw what (one word, zero between setups)
    // outputs
    double *x, *lam, *fval, *soft;
    int *exitflag, *iter;
    // debugging
    int *trace; int trace_cap;
    long long *prof;   // [N][8] phase cycle sums, or NULL
    const DAQPSettings *st_dev;   // device copy of st (scalar-load friendly)
    int exact_setup;              // 1: M = A R^-1 in the reference's operation order (VALU); 0: MFMA f64
    int shared;                   // 1: one H, A for the whole batch (daqp_batch_setup_shared): Mblk, Rinv, scaling hold ONE problem's factors
    DAQPSettings st;
    // regularising re-runs of k_setup (utils.c:356-377): only the problems flagged DAQP_NEEDS_SHIFT, with H + hshift[q] on the diagonal
    // (1), or the one setup pass of an LP batch (2): b.H is ONE identity matrix -- the reference's Rinv == RinvD == NULL
    // branches are its RinvD branches with RinvD = 1 (utils.c:455-468,478-481, daqp.c:119-134), every direction proximal
    int prox_pass;
    const double *hshift;         // [N]
    int *prox_mask;               // [N][n] coordinates that carry the shift (all of them for a dense H)
    // workgroup-per-problem solve kernel (wg_kernel.hip.h): persistent workgroups, scratch per WORKGROUP
    int *wg_counter;              // next problem to take
    double *wg_rowc, *wg_rowcT;   // [grid][cap][ldr] active rows, [grid][n][wg_capT] the same transposed
    int *fallback;                // [N] 1: the working set outgrew the LDS-resident L -- the one-wave kernel solves this problem
    int wg_capL, wg_capT;
    int wg_r0;                    // tiered launch of the workgroup kernel: rows of the inverse factor in LDS (0: no such launch)
    int *img_ho;                  // image kernels: problems handed over by the current launch (one counter per batch; the host reads it one launch late), or null
    int img_rows;                 // image kernels: the working-set rows they hold at all (their L in LDS); beyond: hand-over to the full-register kernel of the shape
    int img_cache;                // image kernels (k_ldp_reg<..., IMG != 0>): rows of the active-row cache kept in LDS; slots from here on live in rowc_g (reg_rows - img_cache rows
                                  // per problem), see RWave::cache_slots.  reg_rows: the working-set rows such a kernel holds at all (its L in LDS)
    int reg_rows;                 // k_ldp_reg<2,32,*>: working-set rows it may hold (64 = one lane per row; below cap: an add beyond them flags the problem in
                                  // `fallback` and leaves it untouched for k_ldp -- n = 64 can reach n + 1 = 65 rows; DAQP_AMD_REG_ROWS lowers it for the tests)
    // settings->time_limit (daqp.c:95-103): per-problem start stamp of the current daqp_batch_solve (null: not armed) and the
    // period of the device's constant clock (s_memrealtime) in seconds
    unsigned long long *tstart;
    double tick_s;
};
// internal setup flag: the Hessian is numerically singular and eps_prox != 0 -- the host re-runs the setup with a shifted
// diagonal (never leaves the library: it ends as 1 or DAQP_EXIT_NONCONVEX)
#define DAQP_NEEDS_SHIFT (-100)
// internal setup flag between k_setup_blk (setup_blk.hip.h) and the launch of k_setup_fast right behind it: this problem's factorisation
// is left to the ordered kernel (never leaves the library either); kSetupOnlyMarked: mask bit of that launch -- only such problems
#define DAQP_NEEDS_ORDERED (-102)
constexpr int kSetupOnlyMarked = 1 << 30;
// internal setup flag while the proximal driver runs: this problem sits out the launch (any negative flag does that)
#define DAQP_PROX_SKIP (-101)

__host__ __device__ inline int round_up(int a, int b) { return (a + b - 1) / b * b; }
// index of problem q's factors (M, R^-1, scaling): its own, or the single shared set
__device__ __forceinline__ size_t qf(const BatchDev &b, int q) { return b.shared ? (size_t)0 : (size_t)q; }

} // namespace daqp_amd
