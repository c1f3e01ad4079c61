/* tests/c/boundary_caller.c -- a plain C caller of the drop-in boundary, in the shape of the reference's own C users
 * (docs/docs/c.md:18-71 and .github/benchmarks/maros_meszaros_runner.c:132-180 of the reference): a brace-initialised
 * DAQPProblem, daqp_quadprog, then setup_daqp / daqp_solve / daqp_update_ldp / free_*.
 *
 * Built by tests/test_c_boundary.py with
 *     gcc -std=c11 -Iinclude tests/c/boundary_caller.c -Ldaqp_amd/lib -ldaqp_amd
 * i.e. against include/daqp_amd.h and libdaqp_amd.so exactly as a C project would.  The _Static_asserts pin the struct
 * ABI of the reference (sizeof 80 / 120 / 64 / 288 and the offsets bindings rely on) at COMPILE time.
 *
 * usage: boundary_caller <problem.bin>      (format: see read_problem)      prints results as hex floats
 */
#include <stddef.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include "daqp_amd.h"

_Static_assert(sizeof(DAQPProblem) == 80, "DAQPProblem: reference types.h:14-50");
_Static_assert(sizeof(DAQPSettings) == 120, "DAQPSettings: reference types.h:52-74");
_Static_assert(sizeof(DAQPResult) == 64, "DAQPResult: reference api.h:15-27");
_Static_assert(sizeof(DAQPWorkspace) == 288, "DAQPWorkspace: reference types.h:187-264");
_Static_assert(offsetof(DAQPProblem, H) == 16 && offsetof(DAQPProblem, f) == 24 && offsetof(DAQPProblem, A) == 32, "problem");
_Static_assert(offsetof(DAQPProblem, bupper) == 40 && offsetof(DAQPProblem, blower) == 48 && offsetof(DAQPProblem, sense) == 56, "problem");
_Static_assert(offsetof(DAQPProblem, break_points) == 64 && offsetof(DAQPProblem, nh) == 72 && offsetof(DAQPProblem, problem_type) == 76, "problem");
_Static_assert(offsetof(DAQPSettings, cycle_tol) == 40 && offsetof(DAQPSettings, iter_limit) == 44, "settings");
_Static_assert(offsetof(DAQPSettings, fval_bound) == 48 && offsetof(DAQPSettings, time_limit) == 112, "settings");
_Static_assert(offsetof(DAQPResult, fval) == 16 && offsetof(DAQPResult, exitflag) == 32 && offsetof(DAQPResult, iter) == 36, "result");
_Static_assert(offsetof(DAQPResult, nodes) == 40 && offsetof(DAQPResult, solve_time) == 48 && offsetof(DAQPResult, setup_time) == 56, "result");
_Static_assert(offsetof(DAQPWorkspace, n) == 8 && offsetof(DAQPWorkspace, M) == 24 && offsetof(DAQPWorkspace, Rinv) == 48, "workspace");
_Static_assert(offsetof(DAQPWorkspace, v) == 56 && offsetof(DAQPWorkspace, sense) == 64 && offsetof(DAQPWorkspace, scaling) == 72, "workspace");
_Static_assert(offsetof(DAQPWorkspace, x) == 88 && offsetof(DAQPWorkspace, lam_star) == 112 && offsetof(DAQPWorkspace, fval) == 128, "workspace");
_Static_assert(offsetof(DAQPWorkspace, WS) == 176 && offsetof(DAQPWorkspace, n_active) == 184 && offsetof(DAQPWorkspace, sing_ind) == 192, "workspace");
_Static_assert(offsetof(DAQPWorkspace, settings) == 224 && offsetof(DAQPWorkspace, timer) == 272 && offsetof(DAQPWorkspace, Mu) == 280, "workspace");

typedef struct { int n, m, ms, T; double *H, *f, *A, *bu, *bl, *fs; int *sense; } Problem;

static void *xread(FILE *fp, size_t bytes)
{
    void *p = malloc(bytes ? bytes : 1);
    if (bytes && fread(p, 1, bytes, fp) != bytes) { fprintf(stderr, "short read\n"); exit(2); }
    return p;
}
/* int32 n, m, ms, T | H n*n | f n | A (m-ms)*n | bupper m | blower m | int32 sense m | fs T*n   (doubles are IEEE binary64) */
static Problem read_problem(const char *path)
{
    Problem p;
    FILE *fp = fopen(path, "rb");
    if (!fp) { perror(path); exit(2); }
    int hdr[4];
    if (fread(hdr, sizeof(int), 4, fp) != 4) exit(2);
    p.n = hdr[0]; p.m = hdr[1]; p.ms = hdr[2]; p.T = hdr[3];
    const size_t n = (size_t)p.n, m = (size_t)p.m, mA = (size_t)(p.m - p.ms);
    p.H = xread(fp, 8 * n * n); p.f = xread(fp, 8 * n); p.A = xread(fp, 8 * mA * n);
    p.bu = xread(fp, 8 * m); p.bl = xread(fp, 8 * m); p.sense = xread(fp, 4 * m); p.fs = xread(fp, 8 * (size_t)p.T * n);
    fclose(fp);
    return p;
}
static void print_vec(const char *tag, const double *v, int len)
{
    printf("%s", tag);
    for (int i = 0; i < len; ++i) printf(" %a", v[i]);
    printf("\n");
}

int main(int argc, char **argv)
{
    if (argc != 2) { fprintf(stderr, "usage: %s <problem.bin>\n", argv[0]); return 2; }
    Problem p = read_problem(argv[1]);
    double *x = malloc(sizeof(double) * (size_t)p.n), *lam = malloc(sizeof(double) * (size_t)(p.m ? p.m : 1));
    int *sense = malloc(sizeof(int) * (size_t)(p.m ? p.m : 1));
    memcpy(sense, p.sense, sizeof(int) * (size_t)p.m);

    /* ---- one-shot: daqp_quadprog (docs/docs/c.md:18-45) */
    DAQPProblem qp = {p.n, p.m, p.ms, p.H, p.f, p.A, p.bu, p.bl, sense};   /* 9 fields, trailing members zero */
    DAQPSettings settings;
    daqp_default_settings(&settings);
    DAQPResult result;
    memset(&result, 0, sizeof(result));
    result.x = x; result.lam = lam;
    daqp_quadprog(&result, &qp, &settings);
    printf("quadprog exitflag %d iter %d nodes %d\n", result.exitflag, result.iter, result.nodes);
    if (result.exitflag < 0) printf("error %s\n", daqp_amd_last_error());
    printf("fval %a\n", result.fval);
    print_vec("x", x, p.n);
    print_vec("lam", lam, p.m);

    /* ---- persistent workspace: setup_daqp -> daqp_solve -> {daqp_update_ldp(UPDATE_v) -> daqp_solve}* (docs/docs/c.md:49-71) */
    DAQPWorkspace work;
    memset(&work, 0, sizeof(work));   /* "must be zero-initialised or have settings set" */
    double setup_time = 0;
    const int flag = setup_daqp(&qp, &work, &setup_time);
    printf("setup %d\n", flag);
    if (flag >= 0) {
        for (int t = 0; t <= p.T; ++t) {
            if (t > 0) {
                qp.f = p.fs + (size_t)(t - 1) * (size_t)p.n;
                printf("update %d\n", daqp_update_ldp(DAQP_UPDATE_v, &work, &qp));
            }
            daqp_solve(&result, &work);
            printf("solve %d exitflag %d iter %d n_active %d\n", t, result.exitflag, result.iter, work.n_active);
            print_vec("x", x, p.n);
        }
        free_daqp_workspace(&work);
        free_daqp_ldp(&work);
    }
    free(x); free(lam); free(sense);
    return 0;   /* (an ordinary exit: see tests/c/mask_caller.c) */
}
