/*
 * oracle/daqp_oracle.c -- TEST INFRASTRUCTURE ONLY.
 *
 * Scalar CPU restatement of the dense dual active-set QP path of DAQP v0.9.1
 * (setup QP->LDP, daqp_ldp iteration, LDP->QP back-transform).  It is the
 * checker that tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
 * compare the HIP path against.  Nothing under daqp_amd/ may link or call it.
 *
 * Parity status: PINNED.  oracle/pin_oracle.py checks this file against the
 * reference library itself (oracle/_ref, built from /root/reference by
 * oracle/Makefile): with the strict-IEEE reference build the results are
 * bit-identical (x, lam, fval, iter, exitflag) on every case it runs, and the
 * golden vectors under tests/golden/ were produced by the reference.
 *
 * Every routine names the reference lines whose arithmetic (operation ORDER
 * included -- it decides the last ulp) it restates.  Written from scratch:
 * one flat workspace, explicit loops for the reference's recursion.
 *
 * Scope (what the reference does when avi==NULL, bnb==NULL, nh<=1, no equality
 * elimination, SOFT_WEIGHTS off): sense bits ACTIVE(1) LOWER(2) IMMUTABLE(4)
 * SOFT(8); a singular or forcibly regularised Hessian (n_prox>0) goes through
 * the proximal outer loop of daqp_prox.c, and so does an LP (H==NULL: R = I,
 * adaptive smoothing, gradient steps).  Anything else returns
 * ORA_EXIT_UNSUPPORTED (-8).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define ORA_EMPTY (-1)
#define ORA_UNCONSTRAINED (-2)
#define ORA_INF 1e30

/* exit flags: constants.h:42-51 */
#define ORA_EXIT_SOFT_OPTIMAL 2
#define ORA_EXIT_OPTIMAL 1
#define ORA_EXIT_INFEASIBLE (-1)
#define ORA_EXIT_CYCLE (-2)
#define ORA_EXIT_UNBOUNDED (-3)
#define ORA_EXIT_ITERLIMIT (-4)
#define ORA_EXIT_NONCONVEX (-5)
#define ORA_EXIT_OVERDETERMINED (-6)
#define ORA_EXIT_UNSUPPORTED (-8)

/* update masks: constants.h:54-61 */
#define ORA_UPD_RINV 1
#define ORA_UPD_M 2
#define ORA_UPD_V 4
#define ORA_UPD_D 8
#define ORA_UPD_SENSE 16
#define ORA_UPD_UNCONSTRAINED 64
#define ORA_UPD_ELIMINATE 128

/* sense bits: constants.h:64-96 */
#define S_ACTIVE 1
#define S_LOWER 2
#define S_IMMUTABLE 4
#define S_SOFT 8
#define S_BINARY 16

/* same field order/size as DAQPSettings (types.h:52-74) so callers can pass
 * the reference's struct bytes straight through */
typedef struct {
    double primal_tol, dual_tol, zero_tol, pivot_tol, progress_tol;
    int cycle_tol, iter_limit;
    double fval_bound, eps_prox, eta_prox, rho_soft, rel_subopt, abs_subopt;
    double sing_tol, refactor_tol, time_limit;
} ora_settings;

typedef struct {
    int n, m, ms, ns, cap;        /* cap = n+ns+1 (api.c:296-313) */
    ora_settings st;
    /* LDP data */
    double *M;                    /* (m-ms) x n, row-major, rows normalised */
    double *R;                    /* packed upper R^-1, or the diagonal if is_diag */
    int is_diag;
    int is_lp;                    /* H == NULL: Rinv == RinvD == NULL in the reference (R = I); is_diag is set too */
    double *v, *scaling, *dupper, *dlower;
    int *sense;
    /* iterate + factor */
    double *ubuf, *xbuf;          /* the reference's u(=x) and xold buffers */
    double *u, *x;
    double *lamA, *lamB, *lam, *lam_star;
    double *L, *D, *xldl, *zldl;
    int *WS;
    int n_active, reuse_ind, sing_ind, iterations;
    double fval, soft_slack;
    /* proximal outer loop (types.h:228-229; nh counts its outer iterations, daqp_prox.c:34) */
    int n_prox, nh;
    int *prox_mask;
    /* borrowed problem data (what work->qp points at) */
    const double *qH, *qf, *qA, *qbu, *qbl;
    const int *qsense;
    /* optional event trace: +(id+1) add, -(id+1) remove */
    int *trace; int trace_cap, trace_len;
} ora_work;

static int tri(int k) { return k * (k + 1) / 2; }                 /* DAQP_ARSUM */
static int roff(int i, int n) { return ((2 * n - i - 1) * i) / 2; } /* DAQP_R_OFFSET */

/* branch markers (same codes as the GPU kernels' traces): pivot_last swapped, singular direction, refine_active, refactor
 * repair at a KKT point, cycle-guard rebuild */
enum { ORA_TRACE_MARK = 0x40000000, ORA_TRACE_PIVOT = ORA_TRACE_MARK + 1, ORA_TRACE_SINGULAR = ORA_TRACE_MARK + 2, ORA_TRACE_REFINE = ORA_TRACE_MARK + 3,
       ORA_TRACE_REFACTOR = ORA_TRACE_MARK + 4, ORA_TRACE_CYCLE_RESET = ORA_TRACE_MARK + 5 };
static void trace_ev(ora_work *w, int ev)
{
    if (w->trace && w->trace_len < w->trace_cap) w->trace[w->trace_len] = ev;
    w->trace_len++;
}

void ora_default_settings(ora_settings *s) /* api.c:505-527, constants.h:15-29 */
{
    s->primal_tol = 1e-6;  s->dual_tol = 1e-12; s->zero_tol = 1e-11;
    s->pivot_tol = 1e-6;   s->progress_tol = 1e-14;
    s->cycle_tol = 10;     s->iter_limit = 10000; s->fval_bound = ORA_INF;
    s->eps_prox = -1e-6;   s->eta_prox = -1.0;  s->rho_soft = 1e-6;
    s->rel_subopt = 0;     s->abs_subopt = 0;
    s->sing_tol = 3.7e-11; s->refactor_tol = 1e-9; s->time_limit = 0;
}

/* ------------------------------------------------------------------ */
/* workspace lifetime (api.c:296-371)                                  */
/* ------------------------------------------------------------------ */
ora_work *ora_create(int n, int m, int ms, int ns, const ora_settings *st)
{
    ora_work *w = (ora_work *)calloc(1, sizeof(ora_work));
    int cap = n + ns + 1, mA = m - ms;
    w->n = n; w->m = m; w->ms = ms; w->ns = ns; w->cap = cap;
    if (st) w->st = *st; else ora_default_settings(&w->st);
    w->M = (double *)calloc((size_t)(mA > 0 ? mA : 1) * n, sizeof(double));
    w->R = (double *)calloc((size_t)tri(n) + 1, sizeof(double));
    w->v = (double *)calloc(n, sizeof(double));
    w->scaling = (double *)calloc(m + 1, sizeof(double));
    w->dupper = (double *)calloc(m + 1, sizeof(double));
    w->dlower = (double *)calloc(m + 1, sizeof(double));
    w->sense = (int *)calloc(m + 1, sizeof(int));
    w->ubuf = (double *)calloc(n, sizeof(double));
    w->xbuf = (double *)calloc(n, sizeof(double));
    w->u = w->x = w->ubuf;
    w->lamA = (double *)calloc(cap, sizeof(double));
    w->lamB = (double *)calloc(cap, sizeof(double));
    w->lam = w->lamA; w->lam_star = w->lamB;
    w->L = (double *)calloc((size_t)tri(cap) + cap, sizeof(double));
    w->D = (double *)calloc(cap, sizeof(double));
    w->xldl = (double *)calloc(cap, sizeof(double));
    w->zldl = (double *)calloc(cap, sizeof(double));
    w->WS = (int *)calloc(cap, sizeof(int));
    w->prox_mask = (int *)calloc(n + 1, sizeof(int)); /* api.c:322-323 */
    w->nh = 1;
    for (int i = 0; i < ms; i++) w->scaling[i] = 1; /* api.c:346 */
    w->sing_ind = ORA_EMPTY;
    return w;
}

void ora_free(ora_work *w)
{
    if (!w) return;
    free(w->M); free(w->R); free(w->v); free(w->scaling); free(w->dupper);
    free(w->dlower); free(w->sense); free(w->ubuf); free(w->xbuf);
    free(w->lamA); free(w->lamB); free(w->L); free(w->D); free(w->xldl);
    free(w->zldl); free(w->WS); free(w->prox_mask); free(w);
}

void ora_set_trace(ora_work *w, int *buf, int cap) { w->trace = buf; w->trace_cap = cap; w->trace_len = 0; }
int ora_trace_len(const ora_work *w) { return w->trace_len; }

static void reset_ws(ora_work *w) /* daqp.c:142-146 */
{
    w->sing_ind = ORA_EMPTY; w->n_active = 0; w->reuse_ind = 0;
}

/* row of the LDP constraint matrix: general rows live in M, simple-bound rows
 * are rows of the (normalised) R^-1, which start at column id */
static const double *ldp_row(const ora_work *w, int id, int *first_col)
{
    if (id < w->ms) {
        *first_col = id;
        if (w->is_diag) return NULL;              /* unit row (Rinv==NULL in the reference) */
        return w->R + roff(id, w->n);             /* indexable by absolute column */
    }
    *first_col = 0;
    return w->M + (size_t)w->n * (id - w->ms);
}

/* ------------------------------------------------------------------ */
/* LDL' row append / delete (factorization.c)                          */
/* ------------------------------------------------------------------ */
static double dot4(const double *a, const double *b, int len) /* factorization.c:4-15 */
{
    double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
    int i = 0;
    for (; i + 3 < len; i += 4) {
        s0 += a[i] * b[i];         s1 += a[i + 1] * b[i + 1];
        s2 += a[i + 2] * b[i + 2]; s3 += a[i + 3] * b[i + 3];
    }
    for (; i < len; i++) s0 += a[i] * b[i];
    return (s0 + s1) + (s2 + s3);
}

static void ldl_append(ora_work *w, int id) /* factorization.c:21-111 */
{
    const int na = w->n_active, n = w->n;
    const int base = tri(na);
    int c0, ck, ns_act = 0;
    double acc;
    const double *Mi = ldp_row(w, id, &c0);

    w->sing_ind = ORA_EMPTY;
    acc = Mi ? dot4(Mi + c0, Mi + c0, n - c0) : 1.0;
    if (w->sense[id] & S_SOFT) { acc += w->st.rho_soft; ns_act++; }
    w->D[na] = acc;
    if (na == 0) return;

    for (int k = 0; k < na; k++) {               /* Gram column  M_k . M_i */
        int idk = w->WS[k], j;
        const double *Mk = ldp_row(w, idk, &ck);
        if (w->sense[idk] & S_SOFT) ns_act++;
        j = (idk < w->ms) ? (c0 > idk ? c0 : idk) : c0;
        if (Mk == NULL) acc = Mi ? Mi[j] : 0.0;
        else if (Mi == NULL) acc = Mk[j];
        else acc = dot4(Mk + j, Mi + j, n - j);
        w->L[base + k] = acc;
    }
    for (int i = 0, p = 0; i < na; i++) {        /* forward substitution with L */
        acc = w->L[base + i];
        for (int j = 0; j < i; j++) acc -= w->L[p++] * w->L[base + j];
        w->L[base + i] = acc;
        p++;
    }
    acc = w->D[na];                              /* scale by D, Schur complement */
    for (int i = 0; i < na; i++) {
        double t = w->L[base + i];
        w->L[base + i] /= w->D[i];
        acc -= t * w->L[base + i];
    }
    w->D[na] = acc;
    if (acc < w->st.sing_tol || na >= n + ns_act) { w->sing_ind = na; w->D[na] = 0; }
}

static void ldl_delete(ora_work *w, int r) /* factorization.c:112-151 */
{
    const int na = w->n_active;
    if (na == r + 1) return;
    const int nupd = na - r - 1;
    double *wv = &w->zldl[r];
    int dst = tri(r), src = dst + r + 1, cnt = 0;
    for (int i = r + 1; i < na; src++, dst++, i++)
        for (int j = 0; j < i; j++) {
            if (j != r) w->L[dst++] = w->L[src++];
            else wv[cnt++] = w->L[src++];
        }
    /* Gill-Golub-Murray-Saunders C1 rank-one update of the trailing block */
    double alpha = w->D[r];
    int rowp = tri(r) + r;
    for (int j = 0, i = r + 1; j < nupd; j++, i++) {
        double p = wv[j];
        double dbar = w->D[i] + alpha * p * p;
        w->D[i - 1] = dbar;
        double beta = p * alpha / dbar;
        alpha = w->D[i] * alpha / dbar;
        rowp += i;
        for (int q = j + 1, pos = rowp + j; q < nupd; q++) {
            wv[q] -= p * w->L[pos];
            w->L[pos] += beta * wv[q];
            pos += r + q + 1;
        }
    }
}

/* ------------------------------------------------------------------ */
/* working-set edits (auxiliary.c:3-44, 379-396)                       */
/* ------------------------------------------------------------------ */
static int drop_core(ora_work *w, int r) /* auxiliary.c:3-22; returns 1 if the removal left a singular factor */
{
    trace_ev(w, -(w->WS[r] + 1));
    w->sense[w->WS[r]] &= ~S_ACTIVE;
    ldl_delete(w, r);
    w->n_active--;
    for (int i = r; i < w->n_active; i++) { w->WS[i] = w->WS[i + 1]; w->lam[i] = w->lam[i + 1]; }
    if (r < w->reuse_ind) w->reuse_ind = r;
    if (w->n_active > 0 && w->D[w->n_active - 1] < w->st.sing_tol) {
        w->sing_ind = w->n_active - 1;
        w->D[w->n_active - 1] = 0;
        return 1;
    }
    return 0;
}

static void push_core(ora_work *w, int id, double lam) /* auxiliary.c:27-40 */
{
    trace_ev(w, id + 1);
    w->sense[id] |= S_ACTIVE;
    ldl_append(w, id);
    w->WS[w->n_active] = id;
    w->lam[w->n_active] = lam;
    w->n_active++;
}

/* daqp_pivot_last (auxiliary.c:379-396).  In the reference it recurses through
 * remove_constraint (whose non-singular tail pivots again) and add_constraint
 * (same).  Here: an explicit stack of constraints waiting to be re-inserted.
 * A frame that sees a singular factor after its removal returns without
 * re-inserting, and so does every frame below it -- hence the plain `break`. */
static void pivot_tail(ora_work *w)
{
    int *pend_id = (int *)malloc(sizeof(int) * (w->cap + 1));
    double *pend_lam = (double *)malloc(sizeof(double) * (w->cap + 1));
    int depth = 0;
    for (;;) {
        int r = w->n_active - 2;
        if (w->n_active > 1 && w->D[r] < w->st.pivot_tol && w->D[r] < w->D[w->n_active - 1]) {
            trace_ev(w, ORA_TRACE_PIVOT);
            pend_id[depth] = w->WS[r]; pend_lam[depth] = w->lam[r]; depth++;
            if (drop_core(w, r)) break;
            continue;                             /* the removal's own tail pivot */
        }
        if (depth == 0) break;
        if (w->sing_ind != ORA_EMPTY) break;
        depth--;
        push_core(w, pend_id[depth], pend_lam[depth]);
    }
    free(pend_id); free(pend_lam);
}

static void remove_constraint(ora_work *w, int r) /* auxiliary.c:3-26 */
{
    if (!drop_core(w, r)) pivot_tail(w);
}

static void add_constraint(ora_work *w, int id, double lam)
{
    push_core(w, id, lam);
    pivot_tail(w);
}

/* ------------------------------------------------------------------ */
/* per-iteration kernels (auxiliary.c)                                 */
/* ------------------------------------------------------------------ */
static void solve_csp(ora_work *w) /* auxiliary.c:314-354 */
{
    const int na = w->n_active;
    for (int i = w->reuse_ind, p = tri(w->reuse_ind); i < na; i++) {
        int id = w->WS[i];
        double acc = (w->sense[id] & S_LOWER) ? -w->dlower[id] : -w->dupper[id];
        for (int j = 0; j < i; j++) acc -= w->L[p++] * w->xldl[j];
        p++;
        w->xldl[i] = acc;
    }
    for (int i = w->reuse_ind; i < na; i++) w->zldl[i] = w->xldl[i] / w->D[i];
    int tail = tri(na) - 1;
    for (int i = na - 1; i >= 0; i--) {
        double acc = w->zldl[i];
        int p = tail--;
        for (int j = na - 1; j > i; j--) { acc -= w->lam_star[j] * w->L[p]; p -= j; }
        w->lam_star[i] = acc;
    }
    w->reuse_ind = na;
}

static void singular_direction(ora_work *w) /* auxiliary.c:357-376 */
{
    const int s = w->sing_ind, base = tri(s);
    int tail = base - 1;
    for (int i = s - 1; i >= 0; i--) {
        w->lam_star[i] = -w->L[base + i];
        int p = tail--;
        for (int j = s - 1; j > i; j--) { w->lam_star[i] -= w->lam_star[j] * w->L[p]; p -= j; }
    }
    w->lam_star[s] = 1;
    if (w->sense[w->WS[s]] & S_LOWER)
        for (int i = 0; i <= s; i++) w->lam_star[i] = -w->lam_star[i];
}

static int remove_blocking(ora_work *w) /* auxiliary.c:277-311 */
{
    int pick = ORA_EMPTY;
    double alpha = ORA_INF;
    const double dtol = w->st.dual_tol;
    for (int i = 0; i < w->n_active; i++) {
        int sn = w->sense[w->WS[i]];
        if (sn & S_IMMUTABLE) continue;
        if (sn & S_LOWER) { if (w->lam_star[i] < dtol) continue; }
        else if (w->lam_star[i] > -dtol) continue;
        double cand = (w->sing_ind == ORA_EMPTY)
            ? -w->lam[i] / (w->lam_star[i] - w->lam[i])
            : -w->lam[i] / w->lam_star[i];
        if (cand < alpha) { alpha = cand; pick = i; }
    }
    if (pick == ORA_EMPTY) return 0;
    if (w->sing_ind == ORA_EMPTY)
        for (int i = 0; i < w->n_active; i++) w->lam[i] += alpha * (w->lam_star[i] - w->lam[i]);
    else
        for (int i = 0; i < w->n_active; i++) w->lam[i] += alpha * w->lam_star[i];
    w->sing_ind = ORA_EMPTY;
    remove_constraint(w, pick);
    return 1;
}

static void primal_and_fval(ora_work *w) /* auxiliary.c:46-88 */
{
    const int n = w->n;
    double fv = 0;
    for (int j = 0; j < n; j++) w->u[j] = 0;
    w->soft_slack = 0;
    for (int i = 0; i < w->n_active; i++) {
        int id = w->WS[i], c0;
        const double li = w->lam_star[i];
        const double *row = ldp_row(w, id, &c0);
        if (row == NULL) w->u[id] -= li;
        else if (id < w->ms) for (int j = id; j < n; j++) w->u[j] -= row[j] * li;
        else for (int j = 0; j < n; j++) w->u[j] -= row[j] * li;
        if (w->sense[id] & S_SOFT) fv += w->lam_star[i] * w->lam_star[i];
    }
    fv = fv * w->st.rho_soft;
    w->soft_slack = fv;
    for (int j = 0; j < n; j++) fv += w->u[j] * w->u[j];
    w->fval = fv;
}

/* M*u for the general rows: auxiliary.c:172-198.  Four rows share a pass in the
 * reference; every row is still a k-ordered sum, which is all that matters. */
static double row_dot_u(const double *row, const double *u, int len)
{
    double s = 0;
    for (int k = 0; k < len; k++) s += row[k] * u[k];
    return s;
}

static int add_infeasible(ora_work *w) /* auxiliary.c:89-167 */
{
    const int n = w->n, ms = w->ms, m = w->m;
    const double ep = -w->st.primal_tol;
    double best = 0.0, cand, mu, bound;
    int pick = ORA_EMPTY, upper = 0;
    for (int j = 0; j < m; j++) {
        if (w->sense[j] & (S_ACTIVE + S_IMMUTABLE)) continue;
        if (j < ms) mu = w->is_diag ? w->u[j] : row_dot_u(w->R + roff(j, n) + j, w->u + j, n - j);
        else mu = row_dot_u(w->M + (size_t)n * (j - ms), w->u, n);
        bound = ep * w->scaling[j];
        cand = w->dupper[j] - mu;
        if (cand < best && cand < bound) { pick = j; upper = 1; best = cand; }
        else {
            cand = mu - w->dlower[j];
            if (cand < best && cand < bound) { pick = j; upper = 0; best = cand; }
        }
    }
    if (pick == ORA_EMPTY) return 0;
    if (upper) w->sense[pick] &= ~S_LOWER; else w->sense[pick] |= S_LOWER;
    double *t = w->lam; w->lam = w->lam_star; w->lam_star = t;
    add_constraint(w, pick, upper ? 1.0 : -1.0);
    return 1;
}

static void refine_active(ora_work *w) /* auxiliary.c:498-593 */
{
    const int n = w->n, na = w->n_active;
    w->reuse_ind = 0;
    for (int i = 0; i < na; i++) {
        int id = w->WS[i], c0;
        const double *row = ldp_row(w, id, &c0);
        double mu;
        if (row == NULL) mu = w->u[id];
        else { mu = 0; for (int j = c0; j < n; j++) mu += row[j] * w->u[j]; }
        double d = (w->sense[id] & S_LOWER) ? w->dlower[id] : w->dupper[id];
        w->xldl[i] = mu - d;
        if (w->sense[id] & S_SOFT) w->xldl[i] -= w->st.rho_soft * w->lam_star[i];
    }
    for (int i = 0, p = 0; i < na; i++) {
        double acc = w->xldl[i];
        for (int j = 0; j < i; j++) acc -= w->L[p++] * w->xldl[j];
        p++;
        w->xldl[i] = acc;
    }
    for (int i = 0; i < na; i++) w->zldl[i] = w->xldl[i] / w->D[i];
    int tail = tri(na) - 1;
    for (int i = na - 1; i >= 0; i--) {
        double acc = w->zldl[i];
        int p = tail--;
        for (int j = na - 1; j > i; j--) { acc -= w->xldl[j] * w->L[p]; p -= j; }
        w->xldl[i] = acc;
    }
    for (int i = 0; i < na; i++) w->lam_star[i] += w->xldl[i];
    for (int i = 0; i < na; i++) {
        int id = w->WS[i], c0;
        double dl = w->xldl[i];
        const double *row = ldp_row(w, id, &c0);
        if (row == NULL) w->u[id] -= dl;
        else for (int j = c0; j < n; j++) w->u[j] -= row[j] * dl;
    }
    double fv = w->soft_slack;
    for (int j = 0; j < n; j++) fv += w->u[j] * w->u[j];
    w->fval = fv;
}

static int activate_marked(ora_work *w) /* auxiliary.c:399-479 */
{
    for (int i = 0; i < w->m; i++) {
        if (w->sense[i] & S_ACTIVE)
            add_constraint(w, i, (w->sense[i] & S_LOWER) ? -1.0 : 1.0);
        if (w->sing_ind != ORA_EMPTY) {
            int last = w->WS[w->n_active - 1];
            if (w->sense[last] & S_IMMUTABLE) {
                double resid = 0.0, scale = 1.0;
                singular_direction(w);
                for (int j = 0; j < w->n_active; j++) {
                    int id = w->WS[j];
                    double b = (w->sense[id] & S_LOWER) ? w->dlower[id] : w->dupper[id];
                    double t = w->lam_star[j] * b;
                    resid += t;
                    scale += t < 0 ? -t : t;
                }
                w->sense[last] &= ~S_ACTIVE;
                w->n_active--;
                w->sing_ind = ORA_EMPTY;
                if (w->reuse_ind > w->n_active) w->reuse_ind = w->n_active;
                if (resid <= w->st.primal_tol * scale && resid >= -w->st.primal_tol * scale) continue;
                return ORA_EXIT_OVERDETERMINED;
            }
            int flag = 1;
            for (; i < w->m; i++)
                if (w->sense[i] & S_ACTIVE) {
                    if (w->sense[i] & S_IMMUTABLE) flag = ORA_EXIT_OVERDETERMINED;
                    else w->sense[i] &= ~S_ACTIVE;
                }
            w->n_active--;
            w->sing_ind = ORA_EMPTY;
            return flag;
        }
    }
    return 1;
}

/* ------------------------------------------------------------------ */
/* the dual active-set loop (daqp.c:6-108)                             */
/* ------------------------------------------------------------------ */
static int ldp_loop(ora_work *w)
{
    int flag = ORA_EXIT_ITERLIMIT, it, repaired = 0, stall = 0;
    double best = -1;
    const double fbound = 2 * w->st.fval_bound;
    for (it = 1; it < w->st.iter_limit; ++it) {
        if (w->sing_ind == ORA_EMPTY) {
            solve_csp(w);
            if (remove_blocking(w)) continue;
            primal_and_fval(w);
            if (w->fval > fbound) { flag = ORA_EXIT_INFEASIBLE; break; }
            if (!add_infeasible(w)) {
                double dmin = w->D[0];
                for (int i = 1; i < w->n_active; i++) if (w->D[i] < dmin) dmin = w->D[i];
                if (w->n_active > 2 && repaired != 1 && dmin < w->st.refactor_tol) {
                    repaired = 1;
                    trace_ev(w, ORA_TRACE_REFACTOR);
                    for (int i = 0; i < w->n_active; i++) {
                        if (w->lam[i] >= 0) w->sense[w->WS[i]] &= ~S_LOWER;
                        else w->sense[w->WS[i]] |= S_LOWER;
                    }
                    reset_ws(w);
                    activate_marked(w);
                    continue;
                }
                if (w->n_active > 0 && dmin < w->st.pivot_tol) {
                    trace_ev(w, ORA_TRACE_REFINE);
                    refine_active(w);
                    if (add_infeasible(w)) continue;
                }
                flag = (w->soft_slack > w->st.primal_tol) ? ORA_EXIT_SOFT_OPTIMAL : ORA_EXIT_OPTIMAL;
                break;
            }
            if (w->fval - best < w->st.progress_tol) {
                if (stall++ > w->st.cycle_tol) {
                    if (repaired == 1) { flag = ORA_EXIT_CYCLE; break; }
                    repaired = 1;
                    trace_ev(w, ORA_TRACE_CYCLE_RESET);
                    reset_ws(w);
                    activate_marked(w);
                    stall = 0;
                    best = -1;
                }
            } else { best = w->fval; stall = 0; }
        } else {
            trace_ev(w, ORA_TRACE_SINGULAR);
            singular_direction(w);
            if (!remove_blocking(w)) { flag = ORA_EXIT_INFEASIBLE; break; }
        }
    }
    w->iterations = it;
    return flag;
}

/* ------------------------------------------------------------------ */
/* QP -> LDP (utils.c)                                                 */
/* ------------------------------------------------------------------ */
static int check_bounds(ora_work *w, const double *bu, const double *bl) /* utils.c:546-567 */
{
    int act = 0;
    for (int i = 0; i < w->m; i++) {
        if (w->sense[i] & S_IMMUTABLE) continue;
        double diff = bu[i] - bl[i];
        if (diff < -w->st.primal_tol) return ORA_EXIT_INFEASIBLE;
        else if (diff < w->st.zero_tol && !(w->sense[i] & S_SOFT)) {
            w->sense[i] |= S_ACTIVE + S_IMMUTABLE;
            act = 1;
        }
    }
    return act;
}

/* utils.c:13-20: the proximal shift for a Hessian whose largest |diagonal| is hscale */
static double prox_eps_scaled(const ora_work *w, double hscale)
{
    double eps = w->st.eps_prox;
    if (eps < 0.0) eps = -eps;                 /* negative: automatic mode */
    const double lo = sqrt(w->st.zero_tol) * hscale;
    if (eps > 0.0 && eps < lo) eps = lo;
    return eps;
}

/* utils.c:223-391, unfactored H.  A Hessian that Cholesky finds numerically singular is shifted
 * by eps*I (doubling eps up to 16 times while the shifted factor is still ill-conditioned); a
 * diagonal one is shifted only in its singular coordinates (prox_mask).  n_prox>0 afterwards
 * hands the solve to the proximal outer loop. */
static int factor_hessian(ora_work *w, const double *H)
{
    const int n = w->n;
    const double ztol = w->st.zero_tol;
    const int force = w->st.eps_prox > 0.0;
    double hscale = 0.0, eps = w->st.eps_prox;
    int diag = 1, all = force, tries = 0;
    for (int i = 0; i < n; i++) w->prox_mask[i] = 0;
    w->n_prox = 0;
    for (int i = 0, p = 1; i < n && diag; i++, p += i + 1) {
        double a = H[i * n + i];
        if (a < 0) a = -a;
        if (a > hscale) hscale = a;
        for (int j = 1; j < n - i; j++, p++)
            if (H[p] > ztol || H[p] < -ztol) { diag = 0; break; }
    }
    if (force) {
        if (!diag) {
            hscale = 0.0;
            for (int i = 0; i < n; i++) { double a = H[i * n + i]; if (a < 0.0) a = -a; if (a > hscale) hscale = a; }
        }
        eps = prox_eps_scaled(w, hscale);
        if (eps <= 0.0) return ORA_EXIT_NONCONVEX;
        w->n_prox = n;
        for (int i = 0; i < n; i++) w->prox_mask[i] = 1;
    }
    w->is_diag = diag;
    if (diag) {
        double ftol = ztol;
        if (hscale > 0) ftol = ztol * hscale;
        eps = prox_eps_scaled(w, hscale);
        for (int i = 0; i < n; i++) {
            double h = H[i * n + i];
            if (force || h <= ftol) {
                if (!force) { w->prox_mask[i] = 1; w->n_prox++; }
                h += eps;
            }
            if (h <= ztol) return ORA_EXIT_NONCONVEX;
            h = sqrt(h);
            w->R[i] = 1 / h;
            if (i < w->ms) w->scaling[i] = h;
        }
        return 1;
    }
    double *R = w->R;
    for (;;) {
        for (int i = 0, p = 0; i < n; i++) {          /* pack 1/2 (H + H'), shifted diagonal */
            R[p++] = H[i * n + i] + (all ? eps : 0.0);
            for (int j = i + 1; j < n; j++) R[p++] = 0.5 * (H[i * n + j] + H[j * n + i]);
        }
        double pmin = ORA_INF, pmax = 0.0;
        int ok = 1;
        for (int i = 0, p = 0; i < n; p += n - i, i++) { /* in-place Cholesky, 1/r_ii on the diagonal */
            double dg = R[p];
            for (int k = 0, q = i; k < i; k++, q += n - k) dg -= R[q] * R[q];
            if (dg <= ztol) { ok = 0; break; }
            if (dg < pmin) pmin = dg;
            if (dg > pmax) pmax = dg;
            dg = 1 / sqrt(dg);
            for (int j = 1; j < n - i; j++) {
                for (int k = 0, q = i; k < i; k++, q += n - k) R[p + j] -= R[q] * R[q + j];
                R[p + j] *= dg;
            }
            R[p] = dg;
        }
        /* an already shifted Hessian must clear the stricter sqrt(zero_tol) pivot ratio (utils.c:354-356) */
        if (ok && !(pmin <= ((all && !force) ? sqrt(ztol) : ztol) * pmax)) break;
        if (all) {
            if (eps <= 0 || tries++ >= 16) return ORA_EXIT_NONCONVEX;
            eps *= 2.0;
        } else {
            hscale = 0.0;
            for (int k = 0; k < n; k++) { double a = H[k * n + k]; if (a < 0) a = -a; if (a > hscale) hscale = a; }
            eps = prox_eps_scaled(w, hscale);
            if (eps <= 0) return ORA_EXIT_NONCONVEX;
            all = 1;
            w->n_prox = n;
            for (int k = 0; k < n; k++) w->prox_mask[k] = 1;
        }
    }
    for (int k = 0, p = 0; k < n; k++) {          /* R -> R^-1 in place, row by row */
        int q = p + 1;
        for (int j = k + 1; j < n; j++) R[q++] *= -R[p];
        p++;
        for (int i = k + 1; i < n; i++, p++) {
            R[p] *= R[q++];
            for (int j = 1; j < n - i; j++) R[p + j] -= R[q++] * R[p];
        }
    }
    return 1;
}

/* utils.c:393-432: the shift the factor in w->R was built with, reconstructed at solve time */
static double prox_eps(const ora_work *w)
{
    double scale = 0.0;
    if (w->n_prox == 0 || w->qH == NULL) return 0.0;
    for (int i = 0; i < w->n; i++) { double a = w->qH[i * w->n + i]; if (a < 0.0) a = -a; if (a > scale) scale = a; }
    if (w->is_diag) return prox_eps_scaled(w, scale);
    double rinv = w->R[0];
    if (w->ms > 0) rinv /= w->scaling[0];
    const double recovered = 1.0 / (rinv * rinv) - w->qH[0];
    double eps = prox_eps_scaled(w, scale);
    if (eps <= 0.0) return 0.0;
    while (1.5 * eps < recovered) eps *= 2.0;
    return eps;
}

static void form_v(ora_work *w, const double *f, int mask) /* utils.c:474-497 */
{
    const int n = w->n;
    if (w->is_lp) { for (int i = 0; i < n; i++) w->v[i] = f[i]; return; }
    if (w->is_diag) { for (int i = 0; i < n; i++) w->v[i] = f[i] * w->R[i]; return; }
    int stop = (mask & ORA_UPD_RINV) ? 0 : w->ms;
    int p = tri(n), j;
    for (j = n - 1; j >= stop; j--) {
        for (int i = n - 1; i > j; i--) w->v[i] += w->R[--p] * f[j];
        w->v[j] = w->R[--p] * f[j];
    }
    for (; j >= 0; j--) {
        double fs = f[j] / w->scaling[j];
        for (int i = n - 1; i > j; i--) w->v[i] += w->R[--p] * fs;
        w->v[j] = w->R[--p] * fs;
    }
}

static int normalize_M(ora_work *w) /* utils.c:586-613 */
{
    const int n = w->n;
    for (int i = w->ms, p = 0; i < w->m; i++) {
        double s = 0;
        for (int j = 0; j < n; p++, j++) s += w->M[p] * w->M[p];
        if (s < w->st.zero_tol) {
            w->scaling[i] = 1.0;
            if (w->qbu[i] < -w->st.zero_tol || w->qbl[i] > w->st.zero_tol)
                if (!(w->sense[i] & S_IMMUTABLE) && !(w->sense[i] & S_SOFT)) return ORA_EXIT_INFEASIBLE;
            w->sense[i] = S_IMMUTABLE;
            continue;
        }
        s = 1 / sqrt(s);
        w->scaling[i] = s;
        p -= n;
        for (int j = 0; j < n; j++, p++) w->M[p] *= s;
    }
    return 0;
}

static int form_M(ora_work *w, const double *A, int mask) /* utils.c:434-472 */
{
    const int n = w->n, mA = w->m - w->ms;
    if (!w->is_diag) {
        int stop = (mask & ORA_UPD_RINV) ? n : n - w->ms;
        for (int k = 0, e = n * mA - 1; k < mA; k++, e -= n) {
            int p = tri(n), j;
            for (j = 0; j < stop; ++j) {
                for (int i = 0; i < j; ++i) w->M[e - i] += w->R[--p] * A[e - j];
                w->M[e - j] = w->R[--p] * A[e - j];
            }
            for (; j < n; ++j) {
                double as = A[e - j] / w->scaling[n - j - 1];
                for (int i = 0; i < j; ++i) w->M[e - i] += w->R[--p] * as;
                w->M[e - j] = w->R[--p] * as;
            }
        }
    } else if (w->is_lp) {
        for (int k = 0, p = 0; k < mA; k++)
            for (int i = 0; i < n; i++, p++) w->M[p] = A[p];
    } else {
        for (int k = 0, p = 0; k < mA; k++)
            for (int i = 0; i < n; i++, p++) w->M[p] = A[p] * w->R[i];
    }
    reset_ws(w);
    return normalize_M(w);
}

static void normalize_R(ora_work *w) /* utils.c:569-585 */
{
    if (w->is_diag) return;
    for (int i = 0, p = 0; i < w->ms; i++) {
        double s = 0;
        for (int j = i; j < w->n; j++, p++) s += w->R[p] * w->R[p];
        s = 1 / sqrt(s);
        w->scaling[i] = s;
        p -= (w->n - i);
        for (int j = i; j < w->n; j++, p++) w->R[p] *= s;
    }
}

static void form_d(ora_work *w, const double *bu, const double *bl) /* utils.c:499-544 */
{
    const int n = w->n;
    w->reuse_ind = 0;
    for (int i = 0; i < w->m; i++) {
        w->dupper[i] = bu[i] * w->scaling[i];
        w->dlower[i] = bl[i] * w->scaling[i];
    }
    if (!w->is_diag) {
        for (int i = 0, p = 0; i < w->ms; i++) {
            double s = 0;
            for (int j = i; j < n; j++) s += w->R[p++] * w->v[j];
            w->dupper[i] += s; w->dlower[i] += s;
        }
    } else {
        for (int i = 0; i < w->ms; i++) { w->dupper[i] += w->v[i]; w->dlower[i] += w->v[i]; }
    }
    for (int i = w->ms, p = 0; i < w->m; i++) {
        double s = 0;
        for (int j = 0; j < n; j++) s += w->M[p++] * w->v[j];
        w->dupper[i] += s; w->dlower[i] += s;
    }
}

/* utils.c:618-687: 0 not attempted, 1 computed but infeasible, -2 optimal */
static int try_unconstrained(ora_work *w, int mask)
{
    const int n = w->n;
    if (!(mask & ORA_UPD_UNCONSTRAINED)) return 0;
    if (!(mask & (ORA_UPD_RINV + ORA_UPD_M + ORA_UPD_V + ORA_UPD_D))) return 0;
    if (w->n_prox > 0) return 0;                                  /* utils.c:622 */
    for (int i = 0; i < w->m; i++) if (w->sense[i] & (S_ACTIVE + S_IMMUTABLE)) return 0;
    /* the reference computes x_unc in its xold buffer and swaps pointers so
     * that u (and any warm start in it) survives a negative answer */
    double *xu = w->xbuf;
    int ok = 1;
    const double ptol = w->st.primal_tol;
    if (!w->is_diag) {
        for (int i = 0, p = 0; i < n; i++) {
            double s = 0;
            for (int j = i; j < n; j++) s += w->R[p++] * w->v[j];
            xu[i] = -s;
        }
    } else for (int i = 0; i < n; i++) xu[i] = -w->R[i] * w->v[i];
    for (int i = 0; i < w->ms; i++) {
        w->dupper[i] = w->qbu[i] - xu[i];
        w->dlower[i] = w->qbl[i] - xu[i];
        if (w->dupper[i] < -ptol || w->dlower[i] > ptol) ok = 0;
    }
    for (int i = w->ms, p = 0; i < w->m; i++) {
        double s = 0.0;
        for (int j = 0; j < n; j++) s += w->qA[p++] * xu[j];
        w->dupper[i] = w->qbu[i] - s;
        w->dlower[i] = w->qbl[i] - s;
        if (w->dupper[i] < -ptol || w->dlower[i] > ptol) ok = 0;
    }
    if (ok) {
        reset_ws(w);
        w->sing_ind = ORA_UNCONSTRAINED;
        w->x = w->xbuf;
        return ORA_UNCONSTRAINED;
    }
    return 1;
}

/* eq_elim.c:127-164: would daqp_quadprog's DAQP_UPDATE_eliminate reduce this LDP? */
static int would_eliminate(const ora_work *w)
{
    int neq = 0;
    for (int i = w->ms; i < w->m; i++)
        if ((w->sense[i] & (S_ACTIVE + S_IMMUTABLE + S_SOFT + S_BINARY)) == (S_ACTIVE + S_IMMUTABLE)) neq++;
    if (neq <= 5 || 10 * neq <= w->n) return 0;
    if (w->is_diag && w->m == w->ms + neq) return 0;
    return 1;
}

/* daqp_update_ldp (utils.c:58-221) for the in-scope configuration */
int ora_update(ora_work *w, int mask, const double *H, const double *f, const double *A,
               const double *bu, const double *bl, const int *sense)
{
    int flag, activate = 0, unc;
    if (H) w->qH = H;
    if (f) w->qf = f;
    if (A) w->qA = A;
    if (bu) w->qbu = bu;
    if (bl) w->qbl = bl;
    if (mask & ORA_UPD_SENSE) w->qsense = sense;
    w->sing_ind = ORA_EMPTY;
    if (w->n_prox == 0) w->x = w->ubuf;   /* (the proximal loop's centre lives in x across solves) */
    if (mask & ORA_UPD_SENSE) {
        if (w->qsense == NULL) for (int i = 0; i < w->m; i++) w->sense[i] = 0;
        else {
            for (int i = 0; i < w->m; i++) {
                if (w->qsense[i] & S_BINARY) return ORA_EXIT_UNSUPPORTED;
                w->sense[i] = w->qsense[i];
            }
            activate = 1;
        }
    }
    if (mask & (ORA_UPD_M | ORA_UPD_V | ORA_UPD_D)) {
        flag = check_bounds(w, w->qbu, w->qbl);
        if (flag < 0) return flag;
        if (flag == 1) activate = 1;
    }
    if (mask & ORA_UPD_RINV) {
        flag = factor_hessian(w, w->qH);
        if (flag < 0) return flag;
    }
    if (mask & (ORA_UPD_RINV | ORA_UPD_V)) form_v(w, w->qf, mask);
    unc = try_unconstrained(w, mask);
    if (unc == ORA_UNCONSTRAINED) return 0;
    if ((mask & ORA_UPD_ELIMINATE) && would_eliminate(w)) return ORA_EXIT_UNSUPPORTED;
    if (mask & (ORA_UPD_RINV | ORA_UPD_M)) {
        flag = form_M(w, w->qA, mask);
        if (flag < 0) return flag;
    }
    if (mask & ORA_UPD_RINV) normalize_R(w);
    if (mask & (ORA_UPD_RINV | ORA_UPD_M | ORA_UPD_V | ORA_UPD_D)) {
        if (unc == 1) {
            for (int i = 0; i < w->m; i++) { w->dupper[i] *= w->scaling[i]; w->dlower[i] *= w->scaling[i]; }
            w->reuse_ind = 0;
        } else form_d(w, w->qbu, w->qbl);
    }
    if (activate) {
        reset_ws(w);
        flag = activate_marked(w);
        if (flag < 0) return flag;
    }
    return 0;
}

/* setup_daqp_main + setup_daqp_ldp (api.c:93-209): init_mask 0 = setup_daqp,
 * 64|128 = daqp_quadprog.  Returns 1, or a negative exit flag. */
int ora_setup(ora_work *w, int init_mask, const double *H, const double *f, const double *A,
              const double *bu, const double *bl, const int *sense)
{
    if (f == NULL) return ORA_EXIT_UNSUPPORTED;
    int mask = init_mask | ORA_UPD_M | ORA_UPD_D | ORA_UPD_SENSE | ORA_UPD_V;
    if (H != NULL) mask |= ORA_UPD_RINV;
    else {   /* LP (api.c:183-185): no factor at all, every direction is proximal */
        w->is_lp = 1; w->is_diag = 1; w->n_prox = w->n;
        for (int i = 0; i < w->n; i++) w->R[i] = 1.0;
    }
    int flag = ora_update(w, mask, H, f, A, bu, bl, sense);
    return flag < 0 ? flag : 1;
}

/* ldp2qp_solution (daqp.c:111-139): x = R^-1 (u - v) in place (u aliases x), duals back to the QP's scaling */
static void ldp_to_qp(ora_work *w)
{
    const int n = w->n;
    for (int i = 0; i < n; i++) w->x[i] = w->u[i] - w->v[i];
    if (!w->is_diag) {
        for (int i = 0, p = 0; i < n; i++) {
            w->x[i] *= w->R[p++];
            for (int j = i + 1; j < n; j++) w->x[i] += w->R[p++] * w->x[j];
        }
        for (int i = 0; i < w->ms; i++) w->x[i] /= w->scaling[i];
    } else if (!w->is_lp) for (int i = 0; i < n; i++) w->x[i] *= w->R[i];
    for (int i = 0; i < w->n_active; i++) w->lam_star[i] *= w->scaling[w->WS[i]];
}

/* gradient_step (daqp_prox.c:232-303): an LP iterate that is not at a vertex moves along x - xold until the first
 * constraint blocks, and that constraint joins the working set.  Returns its index or ORA_EMPTY (unbounded). */
static int lp_gradient_step(ora_work *w, const double *xold)
{
    const int n = w->n, m = w->m, ms = w->ms;
    int pick = ORA_EMPTY, lower = 0;
    double amin = ORA_INF;
    for (int j = 0; j < ms; j++) {
        if (w->sense[j] & (S_ACTIVE + S_IMMUTABLE)) continue;
        const double ds = w->x[j] - xold[j];
        if (ds > 0 && w->qbu[j] < ORA_INF && w->qbu[j] - w->x[j] < amin * ds) {
            pick = j; lower = 0; amin = (w->qbu[j] - w->x[j]) / ds;
        } else if (ds < 0 && w->qbl[j] > -ORA_INF && w->qbl[j] - w->x[j] > amin * ds) {
            pick = j; lower = 1; amin = (w->qbl[j] - w->x[j]) / ds;
        }
    }
    for (int j = ms, p = 0; j < m; j++) {
        if (w->sense[j] & (S_ACTIVE + S_IMMUTABLE)) { p += n; continue; }
        double ax = 0, ds = 0;
        for (int k = 0; k < n; k++) { ax += w->M[p] * w->x[k]; ds -= w->M[p++] * xold[k]; }
        ds += ax;
        ax /= w->scaling[j]; ds /= w->scaling[j];
        if (ds > 0 && w->qbu[j] < ORA_INF && w->qbu[j] - ax < ds * amin) {
            pick = j; lower = 0; amin = (w->qbu[j] - ax) / ds;
        } else if (ds < 0 && w->qbl[j] > -ORA_INF && w->qbl[j] - ax > ds * amin) {
            pick = j; lower = 1; amin = (w->qbl[j] - ax) / ds;
        }
    }
    if (pick != ORA_EMPTY) {
        for (int k = 0; k < n; k++) w->x[k] += amin * (w->x[k] - xold[k]);
        if (lower) w->sense[pick] |= S_LOWER; else w->sense[pick] &= ~S_LOWER;
        add_constraint(w, pick, lower ? -1.0 : 1.0);
    }
    return pick;
}

/* daqp_prox (daqp_prox.c:21-221), QP branch: proximal-point iterations x+ = argmin 1/2 x'(H+eps*P)x + (f-eps*P*x)'x
 * over the constraints, each an LDP warm-started from the previous one; P = I for a dense singular H, the
 * singular coordinates of a diagonal one.  Returns the exit flag; iterations = the sum over the inner solves. */
static int prox_loop(ora_work *w)
{
    const int n = w->n;
    const double relax = 1.5;
    const int lp = w->is_lp;
    double eps = lp ? 1.0 : prox_eps(w);
    double eta = w->st.eta_prox;
    int total = 0, relaxed = 0, flag = 0;
    w->nh = 0;
    if (eta < 0.0) {                                   /* automatic tolerance (daqp_prox.c:53-58) */
        eta = 1e-6;
        if (w->st.dual_tol != 1e-12 && 0.1 * w->st.dual_tol < eta) eta = 0.1 * w->st.dual_tol;
    }
    double *xold = (w->x == w->ubuf) ? w->xbuf : w->ubuf;
    while (total < w->st.iter_limit) {
        if (lp) {   /* smoothing weight: grow while the inner LP stalls, shrink otherwise (daqp_prox.c:69-78) */
            if (total > 0) eps *= (w->iterations == 1) ? 10.0 : 0.9;
            if (eps > 1e3) eps = 1e3;
            for (int i = 0; i < n; i++) w->v[i] = w->qf[i] * eps - w->x[i];
        } else {
            if (w->n_prox == n) for (int i = 0; i < n; i++) w->v[i] = w->qf[i] - eps * w->x[i];
            else for (int i = 0; i < n; i++) w->v[i] = w->qf[i] - (w->prox_mask[i] ? eps : 0.0) * w->x[i];
            form_v(w, w->v, 0);
        }
        form_d(w, w->qbu, w->qbl);
        { double *t = xold; xold = w->x; w->x = t; }   /* xold <- x; the inner solve overwrites the other buffer */
        w->u = w->x;
        w->nh++;
        flag = ldp_loop(w);
        total += w->iterations;
        if (flag < 0) break;
        ldp_to_qp(w);
        if (eps == 0) break;
        const double tol = lp ? eta * eps : eta / eps; /* fixed point ||x - xold||_inf < tol */
        int i;
        for (i = 0; i < n; i++) {
            const double df = w->x[i] - xold[i];
            if (df > tol || df < -tol) break;
        }
        if (i == n) {
            if (relaxed && total < w->st.iter_limit) { relaxed = 0; continue; }
            flag = ORA_EXIT_OPTIMAL;
            break;
        }
        if (!lp && w->iterations == 1 && total < w->st.iter_limit) {  /* unchanged working set: over-relax the affine map */
            for (i = 0; i < n; i++) w->x[i] = xold[i] + relax * (w->x[i] - xold[i]);
            relaxed = 1;
        } else relaxed = 0;
        if (lp && w->iterations == 1 && w->n_active != n) {           /* not at a vertex: walk to the next constraint */
            if (lp_gradient_step(w, xold) == ORA_EMPTY) { flag = ORA_EXIT_UNBOUNDED; break; }
        }
    }
    if (total >= w->st.iter_limit) flag = ORA_EXIT_ITERLIMIT;
    if (lp) for (int i = 0; i < w->n_active; i++) w->lam_star[i] /= eps;
    else {
        double pn = 0.0;
        for (int i = 0; i < n; i++) if (w->prox_mask[i]) pn += w->x[i] * w->x[i];
        w->fval += eps * pn;
    }
    w->iterations = total;
    return flag;
}

/* daqp_solve (api.c:8-59) + daqp_extract_result (api.c:455-495) */
int ora_solve(ora_work *w, double *x, double *lam, double *fval, int *iter, double *soft_slack)
{
    const int n = w->n;
    int flag;
    w->nh = 1;
    if (w->sing_ind != ORA_UNCONSTRAINED && w->n_prox > 0) {
        flag = prox_loop(w);
    } else if (w->sing_ind != ORA_UNCONSTRAINED) {
        w->x = w->u = w->ubuf;
        flag = ldp_loop(w);
        if (flag > 0) ldp_to_qp(w);
    } else {
        w->iterations = 1; w->fval = 0; w->soft_slack = 0;
        flag = ORA_EXIT_OPTIMAL;
    }
    for (int i = 0; i < n; i++) x[i] = w->x[i];
    if (lam) {
        for (int i = 0; i < w->m; i++) lam[i] = 0;
        for (int i = 0; i < w->n_active; i++) lam[w->WS[i]] = w->lam_star[i];
    }
    double fv = w->fval;
    if (w->is_lp) { fv = 0; for (int i = 0; i < n; i++) fv += w->qf[i] * w->x[i]; }   /* api.c:479-482 */
    else {
        for (int i = 0; i < n; i++) fv -= w->v[i] * w->v[i];
        fv *= 0.5;
    }
    if (fval) *fval = fv;
    if (iter) *iter = w->iterations;
    if (soft_slack) *soft_slack = w->soft_slack;
    return flag;
}

/* daqp_quadprog (api.c:62-79): returns the exit flag */
int ora_quadprog(int n, int m, int ms, const double *H, const double *f, const double *A,
                 const double *bu, const double *bl, const int *sense, const ora_settings *st,
                 double *x, double *lam, double *fval, int *iter)
{
    int ns = 0;
    if (sense) for (int i = 0; i < m; i++) if (sense[i] & S_SOFT) ns++;
    ora_work *w = ora_create(n, m, ms, ns, st);
    int flag = ora_setup(w, ORA_UPD_UNCONSTRAINED | ORA_UPD_ELIMINATE, H, f, A, bu, bl, sense);
    if (flag >= 0) flag = ora_solve(w, x, lam, fval, iter, NULL);
    ora_free(w);
    return flag;
}

/* batch driver used for parity sweeps and the cpu_baseline 'port' timing:
 * problems are stored back to back (H: n*n, A: (m-ms)*n, bounds/sense/lam: m) */
void ora_quadprog_batch(int N, int n, int m, int ms, const double *H, const double *f, const double *A,
                        const double *bu, const double *bl, const int *sense, const ora_settings *st,
                        double *x, double *lam, double *fval, int *exitflag, int *iter)
{
    const size_t mA = (size_t)(m - ms);
    for (int q = 0; q < N; q++) {
        int it = 0;
        double fv = 0;
        exitflag[q] = ora_quadprog(n, m, ms, H + (size_t)q * n * n, f + (size_t)q * n, A + q * mA * n,
                                   bu + (size_t)q * m, bl + (size_t)q * m, sense ? sense + (size_t)q * m : NULL,
                                   st, x + (size_t)q * n, lam ? lam + (size_t)q * m : NULL, &fv, &it);
        fval[q] = fv; iter[q] = it;
    }
}

/* read-back helpers for tests */
/* daqp_set_primal_start (api.c:636-641): the iterate the proximal loop starts from */
void ora_set_primal_start(ora_work *w, const double *x)
{
    if (w->sing_ind != ORA_UNCONSTRAINED) for (int i = 0; i < w->n; i++) w->x[i] = x[i];
}
int ora_get_prox(const ora_work *w, int *nh, int *mask)
{
    if (nh) *nh = w->nh;
    if (mask) for (int i = 0; i < w->n; i++) mask[i] = w->prox_mask[i];
    return w->n_prox;
}
int ora_get_state(const ora_work *w, int *n_active, int *WS, int *sense, double *D)
{
    *n_active = w->n_active;
    for (int i = 0; i < w->n_active; i++) { WS[i] = w->WS[i]; if (D) D[i] = w->D[i]; }
    if (sense) for (int i = 0; i < w->m; i++) sense[i] = w->sense[i];
    return w->sing_ind;
}
/* the raw iterate (positions 0..cnt-1 of lam / lam_star as the last iteration left them): for diagnosing which decision of a
 * near-degenerate problem another arithmetic takes differently (tools/degenerate_report.py) */
void ora_get_iterate(const ora_work *w, int cnt, double *lam, double *lam_star)
{
    for (int i = 0; i < cnt; i++) { if (lam) lam[i] = w->lam[i]; if (lam_star) lam_star[i] = w->lam_star[i]; }
}
void ora_get_ldp(const ora_work *w, double *M, double *R, double *v, double *dupper, double *dlower, double *scaling)
{
    if (M) memcpy(M, w->M, sizeof(double) * (size_t)(w->m - w->ms) * w->n);
    if (R) memcpy(R, w->R, sizeof(double) * (w->is_diag ? w->n : tri(w->n)));
    if (v) memcpy(v, w->v, sizeof(double) * w->n);
    if (dupper) memcpy(dupper, w->dupper, sizeof(double) * w->m);
    if (dlower) memcpy(dlower, w->dlower, sizeof(double) * w->m);
    if (scaling) memcpy(scaling, w->scaling, sizeof(double) * w->m);
}
