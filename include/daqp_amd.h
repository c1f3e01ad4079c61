/*
 * daqp_amd.h -- C ABI of the MI355X-native batched dual active-set QP path.
 *
 *   minimize   1/2 x'Hx + f'x      subject to   blower <= [x(1:ms); A x] <= bupper
 *
 * Two layers, both exported by libdaqp_amd.so (plain pointers and sizes only):
 *
 *  (1) The single-problem entry points of DAQP v0.9.1, same names, argument
 *      meaning, struct layouts and exit flags, so that a binding written against
 *      the reference's api.h links against this library unchanged.  Each call
 *      is a batch of one on the GPU.
 *        reference include/api.h:29-30   daqp_solve, daqp_quadprog
 *        reference include/api.h:33-34   setup_daqp, setup_daqp_main
 *        reference include/utils.h:11    daqp_update_ldp
 *        reference include/api.h:40-52   allocate_/free_ helpers, daqp_default_settings
 *        reference include/api.h:56-58   daqp_primal_init_active, daqp_dual_init_active
 *        reference include/api.h:35,54, include/daqp.h:12-13   setup_daqp_ldp, daqp_extract_result, daqp_ldp, ldp2qp_solution
 *
 *  (2) The additive batch entry points (daqp_batch_* / daqp_quadprog_batch):
 *      N independent problems of one shape (n, m, ms), stored back to back,
 *      solved by one wavefront each with the working set and LDL' factors in LDS.
 *
 * Only the hot path is implemented: dense convex H (a singular one, and an LP
 * with H == NULL, go through the reference's proximal outer loop, daqp_prox.c);
 * sense bits ACTIVE/LOWER/IMMUTABLE/SOFT.  Binary constraints, hierarchies and AVIs
 * (the reference's bnb/hiqp/avi loops) return DAQP_EXIT_UNSUPPORTED.  There is NO CPU fallback: without a HIP device every
 * entry point fails with DAQP_EXIT_UNSUPPORTED and daqp_amd_last_error() says why.
 */
#ifndef DAQP_AMD_H
#define DAQP_AMD_H

#ifdef __cplusplus
extern "C" {
#endif

typedef double c_float; /* the path computes in fp64 only (reference types.h:8-12 default) */

/* ---- exit flags (reference include/constants.h:42-51) ---- */
#define DAQP_EXIT_SOFT_OPTIMAL 2
#define DAQP_EXIT_OPTIMAL 1
#define DAQP_EXIT_INFEASIBLE -1
#define DAQP_EXIT_CYCLE -2
#define DAQP_EXIT_UNBOUNDED -3
#define DAQP_EXIT_ITERLIMIT -4
#define DAQP_EXIT_NONCONVEX -5
#define DAQP_EXIT_OVERDETERMINED_INITIAL -6
#define DAQP_EXIT_TIMELIMIT -7
#define DAQP_EXIT_UNSUPPORTED -8

/* ---- daqp_update_ldp masks (reference include/constants.h:54-61) ---- */
#define DAQP_UPDATE_Rinv 1
#define DAQP_UPDATE_M 2
#define DAQP_UPDATE_v 4
#define DAQP_UPDATE_d 8
#define DAQP_UPDATE_sense 16
#define DAQP_UPDATE_hierarchy 32
#define DAQP_UPDATE_unconstrained 64
#define DAQP_UPDATE_eliminate 128

/* ---- constraint sense bits (reference include/constants.h:64-96) ---- */
#define DAQP_ACTIVE 1
#define DAQP_LOWER 2
#define DAQP_IMMUTABLE 4
#define DAQP_SOFT 8
#define DAQP_BINARY 16

#define DAQP_EMPTY_IND -1
#define DAQP_UNCONSTRAINED_OPTIMAL -2
#define DAQP_INF ((c_float)1e30)

/* Problem descriptor: layout of reference include/types.h:14-50 (80 bytes). */
typedef struct {
    int n, m, ms;            /* variables, constraints (simple first), simple bounds */
    c_float *H;              /* n x n, row-major */
    c_float *f;              /* n */
    c_float *A;              /* (m-ms) x n, row-major */
    c_float *bupper, *blower; /* m each: [simple; general] */
    int *sense;              /* m, or NULL (= all 0) */
    int *break_points;       /* must be NULL (hierarchies unsupported) */
    int nh;                  /* 0 or 1 */
    int problem_type;        /* must be 0 */
} DAQPProblem;

/* Settings: layout of reference include/types.h:52-74 (120 bytes). */
typedef struct {
    c_float primal_tol, dual_tol, zero_tol, pivot_tol, progress_tol;
    int cycle_tol, iter_limit;
    c_float fval_bound;
    c_float eps_prox, eta_prox; /* shift / stationarity tolerance of the proximal loop (singular H); < 0: automatic */
    c_float rho_soft;
    c_float rel_subopt, abs_subopt;
    c_float sing_tol, refactor_tol;
    c_float time_limit;         /* seconds, 0 = none: checked every 32nd iteration on the device clock, per problem
                                   from the start of its solve (daqp.c:95-103) -> DAQP_EXIT_TIMELIMIT */
} DAQPSettings;

/* Result: layout of reference include/api.h:15-27 (64 bytes). x and lam are caller-owned. */
typedef struct {
    c_float *x, *lam;
    c_float fval, soft_slack;
    int exitflag, iter, nodes;
    c_float solve_time, setup_time;
} DAQPResult;

/* Workspace: field order and offsets of reference include/types.h:187-264
 * (288 bytes), so that sizeof/offsetof match for bindings that mirror it.
 * In this implementation the numerical state lives on the GPU: the array
 * members below are HOST MIRRORS refreshed by setup/solve (NULL where no
 * mirror is kept); `timer` carries the opaque device handle. */
typedef struct DAQPWorkspace {
    DAQPProblem *qp;
    int n, m, ms;
    c_float *M, *dupper, *dlower, *Rinv, *v;   /* READ-ONLY host mirrors of the device-resident LDP (refreshed by setup / update_ldp;
                                                  layouts of the reference: M (m-ms) x n row-major, Rinv packed upper; Rinv is NULL
                                                  for a diagonal H (then RinvD) and for an LP) */
    int *sense;                                /* host mirror, m */
    c_float *scaling, *RinvD;                  /* host mirrors (see above) */
    c_float *x, *xold;                         /* x: host mirror of the last primal solution */
    c_float *lam, *lam_star, *u;               /* lam_star: host mirror (n_active multipliers) */
    c_float fval;
    c_float *L, *D, *xldl, *zldl;              /* NULL */
    int reuse_ind;
    int *WS;                                   /* host mirror of the working set */
    int n_active, iterations, sing_ind;
    int *prox_mask;                            /* NULL */
    int n_prox;                                /* > 0: H was shifted, daqp_solve runs the proximal loop (host mirror) */
    c_float soft_slack;
    DAQPSettings *settings;
    void *bnb;                                 /* NULL */
    int nh;
    int *break_points;
    void *avi, *eq;                            /* NULL */
    void *timer;                               /* opaque: DAQPBatch* of size 1 */
    c_float *Mu;                               /* NULL */
} DAQPWorkspace;

/* ------------------------------------------------------------------ */
/* (1) single-problem drop-in entry points                             */
/* ------------------------------------------------------------------ */
/* LATENCY: each of these is a batch of ONE on the GPU -- two kernels and mapped result slabs: daqp_quadprog 0.13 ms (n = 20, m = 40) to
   0.25 ms (n = 50, m = 150), daqp_update_ldp(UPDATE_v) + daqp_solve on a kept workspace 0.05-0.06 ms (profiles/r05zzzz_latency_one.txt), where
   the reference needs ~0.02 ms on one core for a small problem.
   STALE MIRRORS: daqp_update_ldp(UPDATE_v | UPDATE_d) on a kept workspace is checked on the host and applied by the NEXT daqp_solve's launch; until
   that solve returns, work->v, work->dupper and work->dlower (read-only host copies for bindings that look at them) still hold the previous LDP.  They exist so that bindings written against the reference's api.h
   link and behave; a caller that loops over many QPs of one shape should hand them over at once: section (2), daqp_quadprog_batch. */
void daqp_quadprog(DAQPResult *res, DAQPProblem *qp, DAQPSettings *settings); /* api.c:62-79 */
void daqp_solve(DAQPResult *res, DAQPWorkspace *work);                         /* api.c:8-59 */
int setup_daqp(DAQPProblem *qp, DAQPWorkspace *work, c_float *setup_time);     /* api.c:88-90 */
int setup_daqp_main(DAQPProblem *qp, DAQPWorkspace *work, c_float *setup_time, int init_mask); /* api.c:93-160 */
int daqp_update_ldp(const int mask, DAQPWorkspace *work, DAQPProblem *qp);     /* utils.c:58-221 */
void daqp_default_settings(DAQPSettings *settings);                            /* api.c:505-527 */
void allocate_daqp_settings(DAQPWorkspace *work);                              /* api.c:277-282 */
void free_daqp_workspace(DAQPWorkspace *work);                                 /* api.c:393-420 (the device side of a freed workspace is PARKED for the next one of its shape, up to 8: daqp_amd_release_pool() really frees, DAQP_AMD_NO_POOL=1 never parks) */
void free_daqp_ldp(DAQPWorkspace *work);                                       /* api.c:243-275 */
void daqp_primal_init_active(DAQPProblem *qp, c_float *x);                     /* api.c:579-616 */
void daqp_dual_init_active(DAQPProblem *qp, c_float *lam);                     /* api.c:620-633 */
void daqp_set_primal_start(DAQPWorkspace *work, c_float *x);                   /* api.c:636-641 (first centre of the proximal loop) */
void allocate_daqp_workspace(DAQPWorkspace *work, int n, int ns);                /* api.h:41 (records n; state is created by setup_daqp) */
void allocate_daqp_ldp(DAQPWorkspace *work, int n, int m, int ms, int alloc_R, int alloc_v);   /* api.h:42 (records n, m, ms) */
int daqp_first_violating(c_float *x, c_float *A, c_float *bu, c_float *bl, int n, int m, int ms, c_float tol);   /* api.c:562-574, host-only */
void daqp_minrep(int *is_redundant, c_float *A, c_float *b, int n, int m, int ms); /* api.h:55: outside the path (link stub: is_redundant untouched, daqp_amd_last_error() says so) */
int setup_daqp_ldp(DAQPWorkspace *work, DAQPProblem *qp, const int init_mask);      /* api.c:161-209 (api.h:35) */
int daqp_ldp(DAQPWorkspace *work);                                                  /* daqp.c:6-108 (daqp.h:12): iterate from the workspace's state; returns the exit flag */
void ldp2qp_solution(DAQPWorkspace *work);                                          /* daqp.c:111-139 (daqp.h:13): done on the device by daqp_ldp -- a no-op kept for callers of the pair */
void daqp_extract_result(DAQPResult *res, DAQPWorkspace *work);                     /* api.c:455-495 (api.h:54): x, lam (from WS / lam_star / n_active), fval, iter, soft_slack from the workspace's fields */

/* ------------------------------------------------------------------ */
/* (2) batch entry points (additive; not in the reference)             */
/* ------------------------------------------------------------------ */
#define DAQP_MEM_HOST 0    /* pointers are host memory: the library stages them over PCIe */
#define DAQP_MEM_DEVICE 1  /* pointers are device memory on the batch's GPU: used in place */

/* N problems of one shape, each array the N per-problem arrays back to back
 * (H: N*n*n, f: N*n, A: N*(m-ms)*n, bupper/blower/sense: N*m).  sense may be NULL. */
typedef struct {
    int N, n, m, ms;
    const c_float *H, *f, *A, *bupper, *blower;
    const int *sense;
    int memory; /* DAQP_MEM_HOST or DAQP_MEM_DEVICE, applies to every pointer above */
} DAQPBatchProblem;

/* Per-problem outputs, back to back (x: N*n, lam: N*m, others: N).  Any pointer may be NULL. */
typedef struct {
    c_float *x, *lam, *fval, *soft_slack;
    int *exitflag, *iter;
    int memory;
    c_float setup_time, solve_time; /* host wall clock of the last call, seconds (includes device sync) */
} DAQPBatchResult;

typedef struct DAQPBatch DAQPBatch; /* device-resident workspaces of N problems */

/* Allocate device workspaces for N problems of shape (n, m, ms) with at most ns_max soft
 * constraints each, on HIP device `device` (-1: current).  settings NULL = defaults.
 * Returns 0 or a negative exit flag. */
int daqp_batch_create(DAQPBatch **out, int N, int n, int m, int ms, int ns_max,
                      const DAQPSettings *settings, int device);
void daqp_batch_free(DAQPBatch *b);
/* A freed batch of ONE problem (what daqp_quadprog and setup_daqp / free_daqp_workspace create and free per call -- reference
 * api.c:61-104) is parked and handed to the next daqp_batch_create of the same shape, device and environment switches, so that
 * repeated single solves do not pay ~40 hipMalloc / hipFree each.  At most 8 are kept; this releases them all now.
 * Environment DAQP_AMD_NO_POOL=1 switches the parking off. */
void daqp_amd_release_pool(void);
/* waits for everything this library has in flight on every device it touched (frees nothing; any number of calls).  Registered with atexit() at
   the first batch / workspace creation, so that a process that exits right after its last solve does not meet the HIP runtime's own teardown with
   commands still retiring (INTEGRATION.md "Process exit"); DAQP_AMD_NO_EXIT_SYNC=1 leaves that to the host. */
void daqp_amd_shutdown(void);
/* hipStream_t the batch launches on (NULL: the legacy default stream). */
void daqp_batch_set_stream(DAQPBatch *b, void *hip_stream);
void daqp_batch_set_settings(DAQPBatch *b, const DAQPSettings *settings);
/* exact != 0: form M = A R^-1 in the reference's operation order (VALU; the whole LDP is then bit-identical
 * to the reference's strict-IEEE build).  0 (default, or env DAQP_AMD_EXACT unset): MFMA f64 matrix cores,
 * M equal to ~1e-16 relative; active sets/iterations identical, x within 1e-9 (tests/test_gpu_fast_mode.py). */
void daqp_batch_set_exact(DAQPBatch *b, int exact);

/* setup_daqp_main for every problem (QP -> LDP: Cholesky, R^-1, M = A R^-1, v, d, initial working
 * set).  init_mask 0 = setup_daqp, DAQP_UPDATE_unconstrained = the daqp_quadprog variant.
 * Returns 0 when launched; per-problem flags (1 ok, <0 exit flag) via daqp_batch_setup_flags. */
int daqp_batch_setup(DAQPBatch *b, const DAQPBatchProblem *p, int init_mask);
/* Shared-structure batches (condensed MPC; SURVEY.md 8f rank 3, docs/docs/linearmpc.md:18-21 of the reference): p->H is
 * ONE n x n matrix and p->A ONE (m-ms) x n matrix for all N problems; f, bupper, blower (and sense) are per problem.
 * This is the reference's MPC usage, batched: setup_daqp once (api.c:88-151, open bounds), then for every problem
 * daqp_update_ldp(DAQP_UPDATE_v|DAQP_UPDATE_d) (utils.c:58-221) with its f and bounds -- bit for bit.  The factorisation
 * runs once and every solve reads one shared image of M.  Follow with daqp_batch_solve / daqp_batch_update as usual. */
int daqp_batch_setup_shared(DAQPBatch *b, const DAQPBatchProblem *p, int init_mask);
/* daqp_update_ldp (utils.c:58-221) for every problem, ANY mask of DAQP_UPDATE_Rinv | M | v | d | sense -- each bit's step on its
 * own, as the reference runs them and as its bindings send them (daqp.pyx:513-571 builds the mask field by field):
 *   v, d          new f and / or bounds on the kept factors and working sets: the warm path (applied by the next solve launch)
 *   sense         the caller's sense replaces the workspace's (p->sense == NULL: zeros) and, if one was given, the working sets are
 *                 rebuilt from its ACTIVE bits (utils.c:84-91,199-211)
 *   M             new A on the kept R^-1 (its rows < ms stay normalised: utils.c:447-452); working sets emptied (utils.c:470);
 *                 the workspace's sense -- the ACTIVE bits of the last solve included -- is kept unless the sense bit is set too
 *   Rinv          new H: factor, then v, M, the normalisations and d follow as in a setup (utils.c:122,135,142,150); sense as for M
 *   all five      a re-setup (the setup kernels; the iterate of a proximal problem is kept, api.c:318 does not run)
 * DAQP_UPDATE_unconstrained / _eliminate may ride along as in setup_daqp_main.  Arrays of `p` that the mask's steps do not read may
 * be NULL; one that they read and that is NULL stays as the batch has it (device-resident arrays were adopted, not copied: they must
 * still be valid).  Per-problem outcome: daqp_batch_setup_flags (1, or the flag daqp_update_ldp would have returned: -1 for crossed
 * bounds -- that problem's LDP is then untouched and its next solve reports the -1 instead of solving -- -5, ...).
 * Masks with the Rinv or M bit but not all five run the generic one-wave setup kernel in the reference's operation order in both
 * arithmetic modes; they are not available after daqp_batch_setup_shared (one H and A for the whole batch: set it up again). */
int daqp_batch_update(DAQPBatch *b, int mask, const DAQPBatchProblem *p);
/* daqp_solve for every problem: dual active-set iteration from the current working sets, then
 * x, lam, fval, exitflag, iter.  Blocks until results are in `r` unless r->memory is DEVICE
 * (then they are ordered on the batch's stream). */
int daqp_batch_solve(DAQPBatch *b, DAQPBatchResult *r);
/* Singular Hessians (daqp_prox.c:21-221, utils.c:223-432): with eps_prox != 0 (default -1e-6:
 * automatic) a problem whose Hessian Cholesky finds numerically singular is factorised from H + eps*I (eps doubling
 * while still ill-conditioned; a diagonal H is shifted in its singular coordinates only), and daqp_batch_solve runs the
 * reference's proximal-point outer loop for it: inner LDPs warm-started from each other until ||x - x_old||_inf <
 * eta_prox/eps.  iter is the sum over the inner solves.  eps_prox > 0 forces the shift for every problem.
 * daqp_batch_set_primal_start: api.c:636-641 for every problem (x: N*n), the centre of the first outer iteration.
 * daqp_batch_prox_info: n_prox per problem (types.h:229), outer iterations of the last solve, eps; returns the number
 * of proximal problems.  An LP batch: p->H == NULL in daqp_batch_setup (every problem of the batch; R = I, adaptive
 * smoothing weight, gradient steps, exit flag -3 for an unbounded one: daqp_prox.c LP branch). */
int daqp_batch_set_primal_start(DAQPBatch *b, const c_float *x, int memory);
int daqp_batch_prox_info(DAQPBatch *b, int *n_prox_host, int *outer_host, c_float *eps_host);
/* copy out per-problem setup flags (host int[N]) */
int daqp_batch_setup_flags(DAQPBatch *b, int *flags_host);
/* copy out working sets: n_active (host int[N]) and WS (host int[N*(n+ns_max+1)], -1 padded); either may be NULL */
int daqp_batch_working_sets(DAQPBatch *b, int *n_active_host, int *ws_host);
/* one-shot: create + setup(DAQP_UPDATE_unconstrained) + solve + free == N x daqp_quadprog */
int daqp_quadprog_batch(DAQPBatchResult *r, const DAQPBatchProblem *p, const DAQPSettings *settings);

/* ---- several GPUs of this host (independent problems, no exchange step) -------------------------------------------
 * A DAQPMultiBatch is G device-resident shards -- each an ordinary DAQPBatch on its own device and HIP stream, driven by its own
 * persistent host thread -- over which ONE batch of N problems is dealt: problem k lives on shard k mod G (interleaved, so that the
 * spread of iteration counts averages out).  Factors, working sets and iterates stay on the devices between calls, so the
 * reference's setup_daqp -> daqp_solve -> {daqp_update_ldp -> daqp_solve}* sequence runs over G devices from one C process.
 * devices == NULL: 0 .. n_devices-1; n_devices <= 0: every visible device (a list, if passed, is then ignored).  A device may be
 * listed more than once (its shards share it).  G = min(n_devices, N).
 *   daqp_batch_setup_multi / update_multi / solve_multi take ONE host-resident batch in the caller's order (memory must be
 *     DAQP_MEM_HOST): every shard gathers its problems through pinned buffers, chunk by chunk, two deep, into its own device slots;
 *     results come back the same way, each to its own index.  update_multi: any mask of daqp_batch_update (the arrays its steps read;
 *     a full re-setup needs every array).
 *   daqp_batch_*_multi_shards take G descriptors, ps[g] / rs[g] = shard g's problems / results back to back (N = that shard's
 *     size: daqp_batch_multi_shard), host-resident or resident ON THAT SHARD'S DEVICE (used in place); the shards run side by side.
 *   daqp_batch_multi_shard hands out shard g's DAQPBatch (for the inspection calls above: setup flags, working sets, kernel times).
 * Every call returns when all shards have finished it. */
typedef struct DAQPMultiBatch DAQPMultiBatch;
int daqp_batch_create_multi(DAQPMultiBatch **out, int N, int n, int m, int ms, int ns_max, const DAQPSettings *settings,
                            const int *devices, int n_devices);
void daqp_batch_free_multi(DAQPMultiBatch *mb);
int daqp_batch_multi_shards(const DAQPMultiBatch *mb);                                           /* G */
DAQPBatch *daqp_batch_multi_shard(DAQPMultiBatch *mb, int g, int *shard_N, int *device);        /* shard g; either out-pointer may be NULL */
int daqp_batch_setup_multi(DAQPMultiBatch *mb, const DAQPBatchProblem *p, int init_mask);
int daqp_batch_update_multi(DAQPMultiBatch *mb, int mask, const DAQPBatchProblem *p);
int daqp_batch_solve_multi(DAQPMultiBatch *mb, DAQPBatchResult *r);
int daqp_batch_setup_multi_shards(DAQPMultiBatch *mb, const DAQPBatchProblem *ps /* [G] */, int init_mask);
int daqp_batch_update_multi_shards(DAQPMultiBatch *mb, int mask, const DAQPBatchProblem *ps /* [G] */);
int daqp_batch_solve_multi_shards(DAQPMultiBatch *mb, DAQPBatchResult *rs /* [G] */);
/* one-shot on top of it: create + setup(DAQP_UPDATE_unconstrained) + solve + free == N x daqp_quadprog over the listed devices;
 * problems and results host-resident; setup_time / solve_time: the slowest shard's */
int daqp_quadprog_batch_multi(DAQPBatchResult *r, const DAQPBatchProblem *p, const DAQPSettings *settings,
                              const int *devices, int n_devices);

/* device-side timing of the last setup / solve launches (HIP events on the batch's stream), ms */
int daqp_batch_kernel_ms(DAQPBatch *b, float *setup_ms, float *solve_ms);
/* Default arithmetic only: problems that the FIRST solve after a daqp_batch_setup (or after a daqp_batch_update with every bit set)
   declares infeasible are set up and solved again in the reference's own arithmetic, and that result (exit flag, iter, lam, stored
   iterate) is the one reported -- "infeasible" is decided by comparing rounding noise with dual_tol (auxiliary.c:284-287) and only the
   reference's arithmetic reproduces the reference's noise.  The second pass reads the setup's inputs again at solve time.  Host-
   resident inputs were staged into the batch's own buffers: nothing to watch.  DEVICE-resident inputs were adopted, not copied: they
   must still hold what the setup read when the solve runs -- a caller that recycles those buffers between setup and solve switches
   the pass off for that batch, daqp_batch_set_recheck(b, 0) (or DAQP_AMD_RECHECK_DEVICE=0 for the process; DAQP_AMD_NO_RECHECK=1
   switches it off for every batch).  daqp_batch_rechecked: how many problems of the last daqp_batch_solve took the second pass
   (-1: there were some and no memory for the companion batch -- the first pass's verdicts stand); daqp_batch_recheck_ms: its
   device time, which is part of the solve time daqp_batch_kernel_ms reports.  A WARM solve (after an update of f / bounds) that ends
   INFEASIBLE has no such pass: its starting point already carries the default arithmetic's rounding.  THE BOUND: the exit flag is the
   reference's; the iteration at which the verdict falls is within TWO of the reference's (tests/test_gpu_golden.py asserts it per problem).
   Which comparison it is (tools/warm_infeasible_trace.py, profiles/r06_warm_infeasible_traces.txt): the event traces of the two arithmetics are
   identical up to a singular-direction step (daqp.c:86-93); there the blocking test of auxiliary.c:284-287 compares components of the
   singular direction that are zero in exact arithmetic -- rounding noise of the size of dual_tol = 1e-12 -- with dual_tol, and one side
   removes one more row (and looks again) where the other already reports -1.  Not the fval bound of daqp.c:20-23.  DAQP_AMD_EXACT=1
   reproduces the reference's trace event for event. */
int daqp_batch_rechecked(const DAQPBatch *b);
void daqp_batch_set_recheck(DAQPBatch *b, int on);
int daqp_batch_recheck_ms(DAQPBatch *b, float *ms);
/* bytes of device memory held by the batch */
unsigned long long daqp_batch_device_bytes(const DAQPBatch *b);

/* ---- test / tuning hooks (used by tests/ and tools/; not needed by a caller of the solver) ---- */
/* per-problem add/remove event trace: `cap` ints per problem (+(id+1) add, -(id+1) remove; the last slot = event count); 0 = off */
int daqp_batch_enable_trace(DAQPBatch *b, int cap);
int daqp_batch_read_trace(DAQPBatch *b, int *host /* N*cap */);
/* cycle counters of the kernels' phases (32 x int64 per problem); on = 0 switches them off */
int daqp_batch_enable_profile(DAQPBatch *b, int on);
int daqp_batch_read_profile(DAQPBatch *b, long long *host /* N*32 */);
/* LDP of problem q as the reference stores it: M (m-ms) x n row-major, R^-1 packed upper, v, dupper, dlower, scaling; any may be NULL */
int daqp_batch_read_ldp(DAQPBatch *b, int q, c_float *M, c_float *R, c_float *v, c_float *dupper, c_float *dlower, c_float *scaling);

/* diagnostics */
const char *daqp_amd_last_error(void);
int daqp_amd_device_count(void);
const char *daqp_amd_version(void);
/* always 0: the 16-problems-per-wavefront solve kernel of rounds 3-4 (slower than the default one) has been removed; kept for callers that asked */
int daqp_amd_has_tiny(void);

#ifdef __cplusplus
}
#endif
#endif /* DAQP_AMD_H */
