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
    int lo, hi, n, m, ms;
    const double *H, *f, *A, *bu, *bl;
    double *x, *lam, *fval;
    int *flag, *iter;
} slice_t;

static void *run_slice(void *arg)
{
    slice_t *s = (slice_t *)arg;
    const size_t n = s->n, m = s->m, mA = s->m - s->ms;
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
    return NULL;
}

/* returns wall seconds, or -1 if the library/symbol could not be loaded */
double ref_batch_run(const char *libpath, int threads, int N, int n, int m, int ms, const double *H, const double *f,
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
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (int t = 0; t < threads; t++) {
        slice_t s = {fn, (int)((long long)N * t / threads), (int)((long long)N * (t + 1) / threads), n, m, ms,
                     H, f, A, bu, bl, x, lam, fval, flag, iter};
        sl[t] = s;
        pthread_create(&tid[t], NULL, run_slice, &sl[t]);
    }
    for (int t = 0; t < threads; t++) pthread_join(tid[t], NULL);
    clock_gettime(CLOCK_MONOTONIC, &t1);
    free(tid); free(sl);
    return (t1.tv_sec - t0.tv_sec) + 1e-9 * (t1.tv_nsec - t0.tv_nsec);
}
