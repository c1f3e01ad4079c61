/*
 * oracle/ref_batch.c -- TEST INFRASTRUCTURE ONLY: times a daqp_quadprog-compatible library on
 * the host cores.  It dlopen()s the library it is given (oracle/_ref/libdaqp_ref.so = the
 * reference itself; any library exporting the reference's daqp_quadprog works), splits the batch
 * into contiguous slices, one pthread per slice, and calls daqp_quadprog one QP at a time --
 * the loop a CPU user of the reference would write.
 * Struct layouts: include/daqp_amd.h (== reference types.h:14-74, api.h:15-27).
 */
#include <dlfcn.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include "../include/daqp_amd.h"

typedef void (*quadprog_fn)(DAQPResult *, DAQPProblem *, DAQPSettings *);

typedef struct {
    quadprog_fn fn;
    int lo, hi, n, m, ms, passes;
    const double *H, *f, *A, *bu, *bl;
    double *x, *lam, *fval;
    int *flag, *iter;
    pthread_barrier_t *bar;
} slice_t;

static void *run_slice(void *arg)
{
    slice_t *s = (slice_t *)arg;
    const size_t n = s->n, m = s->m, mA = s->m - s->ms;
    pthread_barrier_wait(s->bar);               /* the clock starts when every thread exists and stands here */
    for (int pass = 0; pass < s->passes; pass++)
    for (int q = s->lo; q < s->hi; q++) {
        DAQPProblem qp;
        memset(&qp, 0, sizeof(qp));
        qp.n = s->n; qp.m = s->m; qp.ms = s->ms;
        qp.H = (double *)s->H + q * n * n; qp.f = (double *)s->f + q * n; qp.A = (double *)s->A + q * mA * n;
        qp.bupper = (double *)s->bu + q * m; qp.blower = (double *)s->bl + q * m; qp.sense = NULL;
        DAQPResult r;
        memset(&r, 0, sizeof(r));
        r.x = s->x + q * n; r.lam = s->lam + q * m;
        s->fn(&r, &qp, NULL);
        s->fval[q] = r.fval; s->flag[q] = r.exitflag; s->iter[q] = r.iter;
    }
    pthread_barrier_wait(s->bar);
    return NULL;
}

/* `passes` sweeps over the batch (each QP solved `passes` times: same inputs, same outputs) between two barriers: thread
 * creation and joining are outside the clock.  Returns wall seconds of all passes, or -1 if the library/symbol could not be loaded */
double ref_batch_run(const char *libpath, int threads, int passes, int N, int n, int m, int ms, const double *H, const double *f,
                     const double *A, const double *bu, const double *bl, double *x, double *lam, double *fval, int *flag,
                     int *iter)
{
    void *h = dlopen(libpath, RTLD_NOW | RTLD_LOCAL);
    if (!h) return -1;
    quadprog_fn fn = (quadprog_fn)dlsym(h, "daqp_quadprog");
    if (!fn) return -1;
    if (threads < 1) threads = 1;
    if (threads > N) threads = N;
    pthread_t *tid = (pthread_t *)malloc(sizeof(pthread_t) * threads);
    slice_t *sl = (slice_t *)malloc(sizeof(slice_t) * threads);
    struct timespec t0, t1;
    pthread_barrier_t bar;
    if (passes < 1) passes = 1;
    pthread_barrier_init(&bar, NULL, threads + 1);
    for (int t = 0; t < threads; t++) {
        slice_t s = {fn, (int)((long long)N * t / threads), (int)((long long)N * (t + 1) / threads), n, m, ms, passes,
                     H, f, A, bu, bl, x, lam, fval, flag, iter, &bar};
        sl[t] = s;
        pthread_create(&tid[t], NULL, run_slice, &sl[t]);
    }
    pthread_barrier_wait(&bar);
    clock_gettime(CLOCK_MONOTONIC, &t0);
    pthread_barrier_wait(&bar);
    clock_gettime(CLOCK_MONOTONIC, &t1);
    for (int t = 0; t < threads; t++) pthread_join(tid[t], NULL);
    pthread_barrier_destroy(&bar);
    free(tid); free(sl);
    return (t1.tv_sec - t0.tv_sec) + 1e-9 * (t1.tv_nsec - t0.tv_nsec);
}

/* ---- config C5: setup_daqp -> daqp_solve -> {daqp_update_ldp(UPDATE_v) -> daqp_solve} x T per QP on P threads ------------
 * Phase 1 (untimed): every thread sets up and cold-solves the QPs of its slice, keeping the workspaces.  Phase 2 (timed,
 * between two barriers): the T warm steps of every QP of the slice -- no Python in the loop.  fs: [T][N][n] (the walk of f),
 * outputs per step: x [T][N][n], lam [T][N][m], iter / flag [T][N]. */
typedef int (*setup_fn)(DAQPProblem *, DAQPWorkspace *, double *);
typedef void (*solve_fn)(DAQPResult *, DAQPWorkspace *);
typedef int (*update_fn)(const int, DAQPWorkspace *, DAQPProblem *);
typedef void (*freews_fn)(DAQPWorkspace *);

typedef struct {
    setup_fn setup; solve_fn solve; update_fn update; freews_fn free_ws, free_ldp;
    int lo, hi, N, n, m, ms, T, passes;
    const double *H, *f, *A, *bu, *bl, *fs;
    double *x, *lam;
    int *flag, *iter;
    pthread_barrier_t *bar;
    double warm_s;
} warm_slice_t;

static void *run_warm_slice(void *arg)
{
    warm_slice_t *s = (warm_slice_t *)arg;
    const size_t n = s->n, m = s->m, mA = s->m - s->ms, N = s->N;
    const int cnt = s->hi - s->lo;
    DAQPWorkspace *ws = (DAQPWorkspace *)calloc(cnt > 0 ? cnt : 1, sizeof(DAQPWorkspace));
    DAQPProblem *qps = (DAQPProblem *)calloc(cnt > 0 ? cnt : 1, sizeof(DAQPProblem));
    double *x0 = (double *)malloc(sizeof(double) * n), *l0 = (double *)malloc(sizeof(double) * (m ? m : 1));
    for (int k = 0; k < cnt; k++) {
        const size_t q = s->lo + k;
        DAQPProblem *qp = &qps[k];
        qp->n = s->n; qp->m = s->m; qp->ms = s->ms;
        qp->H = (double *)s->H + q * n * n; qp->f = (double *)s->f + q * n; qp->A = (double *)s->A + q * mA * n;
        qp->bupper = (double *)s->bu + q * m; qp->blower = (double *)s->bl + q * m; qp->sense = NULL;
        s->setup(qp, &ws[k], NULL);
        DAQPResult r;
        memset(&r, 0, sizeof(r));
        r.x = x0; r.lam = l0;
        s->solve(&r, &ws[k]);
    }
    pthread_barrier_wait(s->bar);
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    /* pass 0 walks f forward through fs[0..T-1] and records every step's result; further passes (to make the timed phase long
     * enough to be a measurement) walk the same path back -- fs[T-2], ..., fs[0], the original f -- and forth again: every step is
     * still ONE increment of the random walk away from the previous one, the workload per warm solve is the same; their results
     * go to scratch. */
    for (int pass = 0; pass < s->passes; pass++)
    for (int k = 0; k < cnt; k++) {
        const size_t q = s->lo + k;
        for (int j = 0; j < s->T; j++) {
            const int back = pass & 1;
            const int t = back ? s->T - 2 - j : j;             /* t == -1: the original f */
            qps[k].f = t < 0 ? (double *)s->f + q * n : (double *)s->fs + ((size_t)t * N + q) * n;
            s->update(DAQP_UPDATE_v, &ws[k], &qps[k]);
            DAQPResult r;
            memset(&r, 0, sizeof(r));
            if (pass == 0) { r.x = s->x + ((size_t)t * N + q) * n; r.lam = s->lam + ((size_t)t * N + q) * m; }
            else { r.x = x0; r.lam = l0; }
            s->solve(&r, &ws[k]);
            if (pass == 0) { s->flag[(size_t)t * N + q] = r.exitflag; s->iter[(size_t)t * N + q] = r.iter; }
        }
    }
    clock_gettime(CLOCK_MONOTONIC, &t1);
    s->warm_s = (t1.tv_sec - t0.tv_sec) + 1e-9 * (t1.tv_nsec - t0.tv_nsec);
    pthread_barrier_wait(s->bar);
    for (int k = 0; k < cnt; k++) { s->free_ws(&ws[k]); s->free_ldp(&ws[k]); }
    free(ws); free(qps); free(x0); free(l0);
    return NULL;
}

/* returns the wall seconds of the warm phase (first thread in to last thread out), or -1 */
double ref_warm_run(const char *libpath, int threads, int passes, int N, int n, int m, int ms, int T, const double *H, const double *f,
                    const double *A, const double *bu, const double *bl, const double *fs, double *x, double *lam, int *flag,
                    int *iter)
{
    void *h = dlopen(libpath, RTLD_NOW | RTLD_LOCAL);
    if (!h) return -1;
    setup_fn su = (setup_fn)dlsym(h, "setup_daqp");
    solve_fn so = (solve_fn)dlsym(h, "daqp_solve");
    update_fn up = (update_fn)dlsym(h, "daqp_update_ldp");
    freews_fn fw = (freews_fn)dlsym(h, "free_daqp_workspace"), fl = (freews_fn)dlsym(h, "free_daqp_ldp");
    if (!su || !so || !up || !fw || !fl) return -1;
    if (threads < 1) threads = 1;
    if (threads > N) threads = N;
    pthread_t *tid = (pthread_t *)malloc(sizeof(pthread_t) * threads);
    warm_slice_t *sl = (warm_slice_t *)malloc(sizeof(warm_slice_t) * threads);
    pthread_barrier_t bar;
    pthread_barrier_init(&bar, NULL, threads + 1);
    for (int t = 0; t < threads; t++) {
        warm_slice_t s = {su, so, up, fw, fl, (int)((long long)N * t / threads), (int)((long long)N * (t + 1) / threads), N, n, m, ms, T, passes < 1 ? 1 : passes,
                          H, f, A, bu, bl, fs, x, lam, flag, iter, &bar, 0.0};
        sl[t] = s;
        pthread_create(&tid[t], NULL, run_warm_slice, &sl[t]);
    }
    struct timespec t0, t1;
    pthread_barrier_wait(&bar);                 /* every workspace is set up and cold-solved */
    clock_gettime(CLOCK_MONOTONIC, &t0);
    pthread_barrier_wait(&bar);                 /* every thread has finished its warm steps */
    clock_gettime(CLOCK_MONOTONIC, &t1);
    for (int t = 0; t < threads; t++) pthread_join(tid[t], NULL);
    pthread_barrier_destroy(&bar);
    free(tid); free(sl);
    return (t1.tv_sec - t0.tv_sec) + 1e-9 * (t1.tv_nsec - t0.tv_nsec);
}
