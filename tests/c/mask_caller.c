/* tests/c/mask_caller.c -- a plain C caller of daqp_update_ldp with arbitrary masks on a kept workspace, as the reference's C users
 * and its bindings drive it (docs/docs/c.md:49-71; interfaces/daqp-python/daqp.pyx:513-571 builds the mask field by field and passes
 * the ONE DAQPProblem whose pointers it has just replaced):
 *     setup_daqp -> daqp_solve -> { qp.<array> = new; daqp_update_ldp(mask, &work, &qp) -> daqp_solve }*
 * Built by tests/test_gpu_update_masks.py with gcc -std=c11 -Iinclude ... -ldaqp_amd, like tests/c/boundary_caller.c.
 *
 * usage: mask_caller <sequence.bin>   prints update flags and results as hex floats
 * format: int32 n, m, ms, T | H n*n | f n | A (m-ms)*n | bupper m | blower m | int32 sense m |
 *         T x { int32 mask | 6 x { int32 present [| the array] } in the order H f A bupper blower sense }
 */
#define _GNU_SOURCE
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <execinfo.h>
#include <signal.h>
#include <unistd.h>
#include "daqp_amd.h"

/* a crash inside the library is reported with its call stack (the test prints stderr) */
static void on_crash(int sig)
{
    void *bt[48];
    const int k = backtrace(bt, 48);
    dprintf(2, "signal %d, call stack:\n", sig);
    backtrace_symbols_fd(bt, k, 2);
    _exit(128 + sig);
}


static void *xread(FILE *fp, size_t bytes)
{
    void *p = malloc(bytes ? bytes : 1);
    if (bytes && fread(p, 1, bytes, fp) != bytes) { fprintf(stderr, "short read\n"); exit(2); }
    return p;
}
static int iread(FILE *fp)
{
    int v;
    if (fread(&v, sizeof(int), 1, fp) != 1) { fprintf(stderr, "short read\n"); exit(2); }
    return v;
}
static void print_vec(const char *tag, const double *v, int len)
{
    printf("%s", tag);
    for (int i = 0; i < len; ++i) printf(" %a", v[i]);
    printf("\n");
}
static void report(int t, const DAQPResult *r, const DAQPWorkspace *w, int n, int m)
{
    printf("solve %d exitflag %d iter %d n_active %d fval %a\n", t, r->exitflag, r->iter, w->n_active, r->fval);
    printf("ws");
    for (int i = 0; i < w->n_active; ++i) printf(" %d", w->WS[i]);
    printf("\n");
    print_vec("x", r->x, n);
    print_vec("lam", r->lam, m);
}

int main(int argc, char **argv)
{
    signal(SIGSEGV, on_crash); signal(SIGBUS, on_crash); signal(SIGABRT, on_crash);
    setvbuf(stdout, NULL, _IOLBF, 0);      /* what was printed before a crash is not lost in the pipe */
    if (argc != 2) { fprintf(stderr, "usage: %s <sequence.bin>\n", argv[0]); return 2; }
    FILE *fp = fopen(argv[1], "rb");
    if (!fp) { perror(argv[1]); return 2; }
    const int n = iread(fp), m = iread(fp), ms = iread(fp), T = iread(fp);
    const size_t sn = (size_t)n, sm = (size_t)m, mA = (size_t)(m - ms);
    const size_t bytes[6] = {8 * sn * sn, 8 * sn, 8 * mA * sn, 8 * sm, 8 * sm, 4 * sm};
    void *cur[6];
    for (int a = 0; a < 6; ++a) cur[a] = xread(fp, bytes[a]);
    DAQPProblem qp = {n, m, ms, cur[0], cur[1], cur[2], cur[3], cur[4], cur[5]};
    DAQPWorkspace work;
    memset(&work, 0, sizeof(work));
    DAQPResult res;
    memset(&res, 0, sizeof(res));
    res.x = malloc(8 * sn); res.lam = malloc(8 * (sm ? sm : 1));
    const int flag = setup_daqp(&qp, &work, NULL);
    printf("setup %d\n", flag);
    if (flag < 0) { printf("error %s\n", daqp_amd_last_error()); return 0; }
    daqp_solve(&res, &work);
    report(0, &res, &work, n, m);
    for (int t = 1; t <= T; ++t) {
        const int mask = iread(fp);
        for (int a = 0; a < 6; ++a)
            if (iread(fp)) { free(cur[a]); cur[a] = xread(fp, bytes[a]); }
        qp.H = cur[0]; qp.f = cur[1]; qp.A = cur[2]; qp.bupper = cur[3]; qp.blower = cur[4]; qp.sense = cur[5];
        const int u = daqp_update_ldp(mask, &work, &qp);
        printf("update %d mask %d\n", u, mask);
        if (u == DAQP_EXIT_UNSUPPORTED) printf("error %s\n", daqp_amd_last_error());
        daqp_solve(&res, &work);
        report(t, &res, &work, n, m);
    }
    fclose(fp);
    free_daqp_workspace(&work);
    free_daqp_ldp(&work);
    printf("end\n");
    /* an ordinary return from main(): the library's own exit handler (daqp_amd_shutdown, registered at the first workspace) waits for what it still
       has in flight before the HIP runtime tears itself down -- tools/stress_caller.py runs this program thousands of times and counts deaths */
    return 0;
}
