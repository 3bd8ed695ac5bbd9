/*
 * oracle/convex_mpc_oracle.c -- TEST INFRASTRUCTURE ONLY (see convex_mpc_assembly.h header).
 *
 * "ConvexMpc, OSQP branch" of the reference, restated in plain C and driven through the REAL
 * vendored OSQP 0.6.0 (oracle/_ref/libosqp_ref.so, compiled by oracle/Makefile from
 * /root/reference/extern/osqp where it lies).  Follows
 *   MPC_Controller/convex_MPC/mpc_osqp.cc:578-796  ComputeContactForces (qp_solver_name_ == OSQP)
 * including the call sequence  osqp_setup | osqp_update_P_A + osqp_update_lin_cost +
 * osqp_update_bounds ; osqp_solve  (mpc_osqp.cc:757-780), the settings (:705-712), the
 * Eigen sparseView()/triangularView<Upper>() conversion (:696-700,727-729: exact zeros pruned,
 * column-major CSC, upper triangle of P), the acceptance rule (only OSQP_SOLVED, :788) and the
 * sign flip of the result (:789-790).
 *
 * This is the CPU baseline of kind "reference" in bench.py and the parity pin of the tests.
 */
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>

#include "convex_mpc_assembly.h"
#include "osqp.h"

typedef struct {
  MpcModel mdl;
  MpcWork wk;
  int n, m;
  double *P, *q, *l, *u, cone[15];
  /* CSC scratch (capacity for the fully dense case) */
  c_int *Pp, *Pi, *Ap, *Ai;
  double *Px, *Ax;
  OSQPWorkspace *work; /* workspace_ (mpc_osqp.cc:276,550) */
  int structure_mismatch;
  int max_iter; /* 0 = OSQP's default (what the reference runs); tests of the MAX_ITER_REACHED path lower it */
} MpcRef;

void *mpcref_create(double mass, const double *inertia9, int h, double dt, double alpha) {
  if (h < 2 || h > MPC_MAX_H) return 0;
  MpcRef *s = (MpcRef *)calloc(1, sizeof(MpcRef));
  mpc_model_init(&s->mdl, mass, inertia9, h, dt, alpha);
  s->n = 12 * h;
  s->m = 20 * h;
  s->P = (double *)calloc((size_t)s->n * s->n, sizeof(double));
  s->q = (double *)calloc(s->n, sizeof(double));
  s->l = (double *)calloc(s->m, sizeof(double));
  s->u = (double *)calloc(s->m, sizeof(double));
  size_t capP = (size_t)s->n * (s->n + 1) / 2, capA = (size_t)15 * 4 * h;
  s->Pp = (c_int *)calloc(s->n + 1, sizeof(c_int));
  s->Pi = (c_int *)calloc(capP, sizeof(c_int));
  s->Px = (double *)calloc(capP, sizeof(double));
  s->Ap = (c_int *)calloc(s->n + 1, sizeof(c_int));
  s->Ai = (c_int *)calloc(capA, sizeof(c_int));
  s->Ax = (double *)calloc(capA, sizeof(double));
  return s;
}

void mpcref_destroy(void *hd) {
  MpcRef *s = (MpcRef *)hd;
  if (!s) return;
  if (s->work) osqp_cleanup(s->work); /* mpc_osqp.cc:192 */
  free(s->P); free(s->q); free(s->l); free(s->u);
  free(s->Pp); free(s->Pi); free(s->Px); free(s->Ap); free(s->Ai); free(s->Ax);
  free(s);
}

/* Eigen sparseView() + triangularView<Upper>() of the dense P (mpc_osqp.cc:696-697,727-729) and
 * sparseView() of the dense constraint matrix (:699-700): keep entries != 0, CSC. */
static void to_csc(MpcRef *s, c_int *nnzP, c_int *nnzA) {
  const int n = s->n, h = s->mdl.h;
  size_t capP = (size_t)n * (n + 1) / 2, capA = (size_t)15 * 4 * h;
  memset(s->Px, 0, capP * sizeof(double));
  memset(s->Ax, 0, capA * sizeof(double));
  c_int k = 0;
  for (int c = 0; c < n; ++c) {
    s->Pp[c] = k;
    for (int r = 0; r <= c; ++r) {
      double v = s->P[(size_t)r * n + c];
      if (v != 0.0) { s->Pi[k] = r; s->Px[k] = v; ++k; }
    }
  }
  s->Pp[n] = k;
  *nnzP = k;
  k = 0;
  for (int c = 0; c < n; ++c) {
    s->Ap[c] = k;
    const int foot = c / 3, cc = c % 3;
    for (int r = 0; r < 5; ++r) {
      double v = s->cone[r * 3 + cc];
      if (v != 0.0) { s->Ai[k] = foot * 5 + r; s->Ax[k] = v; ++k; }
    }
  }
  s->Ap[n] = k;
  *nnzA = k;
}

/*
 * One ComputeContactForces call.  forces_out[12h] receives -x when status == OSQP_SOLVED
 * (mpc_osqp.cc:788-790); otherwise it is left untouched and the function returns 0 (the reference
 * returns an EMPTY vector, :781-794).  Returns 1 on success.
 * info[8]  = {iter, status_val, status_polish, rho_updates, nnzP, nnzA, structure_mismatch, first_run}
 * dinfo[8] = {pri_res, dua_res, rho (settings->rho after the solve), obj_val, c (cost scaling), 0,0,0}
 */
int mpcref_solve(void *hd, const double *in, double *forces_out, int64_t *info, double *dinfo) {
  MpcRef *s = (MpcRef *)hd;
  mpc_assemble(&s->mdl, in, &s->wk, s->P, s->q, s->cone, s->l, s->u);
  c_int nnzP, nnzA;
  to_csc(s, &nnzP, &nnzA);
  for (int i = 0; i < s->m; ++i) { /* mpc_osqp.cc:720-721 */
    if (s->l[i] < -OSQP_INFTY) s->l[i] = -OSQP_INFTY;
    if (s->u[i] > OSQP_INFTY) s->u[i] = OSQP_INFTY;
  }
  int first = 0;
  if (s->work == 0) { /* mpc_osqp.cc:757-759 */
    OSQPSettings settings;
    osqp_set_default_settings(&settings); /* :705-712 */
    settings.verbose = 0;
    settings.warm_start = 1;
    settings.polish = 1;
    settings.adaptive_rho_interval = 25;
    settings.eps_abs = 1e-3;
    settings.eps_rel = 1e-3;
    if (s->max_iter > 0) settings.max_iter = s->max_iter;
    csc Pm = {nnzP, s->n, s->n, s->Pp, s->Pi, s->Px, -1};
    csc Am = {nnzA, s->m, s->n, s->Ap, s->Ai, s->Ax, -1};
    OSQPData data;
    data.n = s->n;
    data.m = s->m;
    data.P = &Pm;
    data.A = &Am;
    data.q = s->q;
    data.l = s->l;
    data.u = s->u;
    if (osqp_setup(&s->work, &data, &settings) != 0) return 0;
    first = 1;
  } else { /* mpc_osqp.cc:760-778 */
    c_int wP = s->work->data->P->p[s->n], wA = s->work->data->A->p[s->n];
    /* The reference hands OSQP the value array of the NEW sparse matrix while OSQP copies as many
     * values as the FIRST problem had non-zeros: a changed zero pattern is undefined behaviour
     * there.  The oracle flags it (value arrays are zero-padded to full capacity, so no OOB). */
    s->structure_mismatch = (wP != nnzP) || (wA != nnzA);
    osqp_update_P_A(s->work, s->Px, OSQP_NULL, nnzP, s->Ax, OSQP_NULL, nnzA);
    osqp_update_lin_cost(s->work, s->q);
    osqp_update_bounds(s->work, s->l, s->u);
  }
  osqp_solve(s->work); /* :780 (the SIGINT branch cannot trigger here) */
  const OSQPInfo *oi = s->work->info;
  if (info) {
    info[0] = oi->iter; info[1] = oi->status_val; info[2] = oi->status_polish; info[3] = oi->rho_updates;
    info[4] = nnzP; info[5] = nnzA; info[6] = s->structure_mismatch; info[7] = first;
  }
  if (dinfo) {
    dinfo[0] = oi->pri_res; dinfo[1] = oi->dua_res; dinfo[2] = s->work->settings->rho; dinfo[3] = oi->obj_val;
    dinfo[4] = s->work->scaling->c; dinfo[5] = dinfo[6] = dinfo[7] = 0;
  }
  if (oi->status_val != OSQP_SOLVED) return 0; /* :788-794 */
  for (int i = 0; i < s->n; ++i) forces_out[i] = -s->work->solution->x[i];
  return 1;
}

/* OSQP's max_iter setting (the reference leaves the default, 4000): lets the tests reach MAX_ITER_REACHED and the *_INACCURATE
 * statuses of osqp.c:563-568 on ordinary problems. */
void mpcref_set_max_iter(void *hd, int max_iter) {
  MpcRef *s = (MpcRef *)hd;
  s->max_iter = max_iter;
  if (s->work && max_iter > 0) osqp_update_max_iter(s->work, max_iter);
}

/*
 * The EXACT optimum of the QP of one call: what the reference's qpOASES branch (mpc_osqp.cc:797-947, the solver the shipped
 * Python selects, ConvexMPCLocomotion.py:108) returns.  qpOASES is an empty submodule in the reference, so its result -- not its
 * iterations -- is pinned: the QP is strictly convex (alpha I), its optimum is unique, and the vendored OSQP reaches it when it is
 * run to eps_abs = eps_rel = 1e-9 with polish from a cold start (no warm start in that branch, :906-919; swing feet, whose bounds
 * are l = u = 0, come out as exact zeros there and to ~1e-10 here).  forces_out = -x.  Returns the OSQP status value.
 * info[4] = {iter, status_val, status_polish, rho_updates}
 */
int mpcref_solve_exact(void *hd, const double *in, double *forces_out, int64_t *info) {
  MpcRef *s = (MpcRef *)hd;
  mpc_assemble(&s->mdl, in, &s->wk, s->P, s->q, s->cone, s->l, s->u);
  c_int nnzP, nnzA;
  to_csc(s, &nnzP, &nnzA);
  for (int i = 0; i < s->m; ++i) {
    if (s->l[i] < -OSQP_INFTY) s->l[i] = -OSQP_INFTY;
    if (s->u[i] > OSQP_INFTY) s->u[i] = OSQP_INFTY;
  }
  OSQPSettings settings;
  osqp_set_default_settings(&settings);
  settings.verbose = 0;
  settings.warm_start = 0;
  settings.polish = 1;
  settings.polish_refine_iter = 10;
  settings.adaptive_rho_interval = 25;
  settings.eps_abs = 1e-9;
  settings.eps_rel = 1e-9;
  settings.max_iter = 200000;
  csc Pm = {nnzP, s->n, s->n, s->Pp, s->Pi, s->Px, -1};
  csc Am = {nnzA, s->m, s->n, s->Ap, s->Ai, s->Ax, -1};
  OSQPData data;
  data.n = s->n; data.m = s->m; data.P = &Pm; data.A = &Am; data.q = s->q; data.l = s->l; data.u = s->u;
  OSQPWorkspace *w = 0;
  if (osqp_setup(&w, &data, &settings) != 0) return -100;
  osqp_solve(w);
  const int st = (int)w->info->status_val;
  if (info) { info[0] = w->info->iter; info[1] = st; info[2] = w->info->status_polish; info[3] = w->info->rho_updates; }
  for (int i = 0; i < s->n; ++i) forces_out[i] = -w->solution->x[i];
  osqp_cleanup(w);
  return st;
}

/*
 * Experiment: the same QP with its swing-leg variables ELIMINATED -- the rows / columns whose bounds are
 * l = u = 0, exactly the reduction of the reference's qpOASES branch (mpc_osqp.cc:838-856) -- solved by the vendored OSQP with the
 * reference's OSQP settings from a cold start.  Answers "does OSQP on the reduced problem return the forces OSQP returns on the
 * full one?" (tools/swing_elimination.py).  forces_out: the full vector (zeros at swing legs), negated.  Returns the status.
 */
int mpcref_solve_reduced(void *hd, const double *in, double *forces_out, int64_t *info) {
  MpcRef *s = (MpcRef *)hd;
  const int n = s->n, m = s->m, h = s->mdl.h;
  mpc_assemble(&s->mdl, in, &s->wk, s->P, s->q, s->cone, s->l, s->u);
  int *vmap = (int *)malloc(sizeof(int) * n), nr = 0, mr = 0;
  int *feet = (int *)malloc(sizeof(int) * 4 * h), nfeet = 0;
  for (int f = 0; f < 4 * h; ++f) {
    int swing = 1;
    for (int r = 0; r < 5; ++r) if (s->l[5 * f + r] != 0.0 || s->u[5 * f + r] != 0.0) swing = 0;
    if (!swing) { feet[nfeet++] = f; for (int c = 0; c < 3; ++c) vmap[nr++] = 3 * f + c; }
  }
  mr = 5 * nfeet;
  for (int i = 0; i < n; ++i) forces_out[i] = 0.0;
  if (nr == 0) { if (info) { info[0] = 0; info[1] = OSQP_SOLVED; info[2] = 0; info[3] = 0; } free(vmap); free(feet); return OSQP_SOLVED; }
  c_int *Pp = (c_int *)malloc(sizeof(c_int) * (nr + 1)), *Pi = (c_int *)malloc(sizeof(c_int) * (size_t)nr * (nr + 1) / 2);
  c_float *Px = (c_float *)malloc(sizeof(c_float) * (size_t)nr * (nr + 1) / 2);
  c_int *Ap = (c_int *)malloc(sizeof(c_int) * (nr + 1)), *Ai = (c_int *)malloc(sizeof(c_int) * 15 * nfeet);
  c_float *Ax = (c_float *)malloc(sizeof(c_float) * 15 * nfeet);
  c_float *q = (c_float *)malloc(sizeof(c_float) * nr), *l = (c_float *)malloc(sizeof(c_float) * mr), *u = (c_float *)malloc(sizeof(c_float) * mr);
  c_int k = 0;
  for (int c = 0; c < nr; ++c) {
    Pp[c] = k;
    for (int r = 0; r <= c; ++r) { double v = s->P[(size_t)vmap[r] * n + vmap[c]]; if (v != 0.0) { Pi[k] = r; Px[k] = v; ++k; } }
    q[c] = s->q[vmap[c]];
  }
  Pp[nr] = k;
  const c_int nnzP = k;
  k = 0;
  for (int c = 0; c < nr; ++c) {
    Ap[c] = k;
    const int foot = c / 3, cc = c % 3;
    for (int r = 0; r < 5; ++r) { double v = s->cone[r * 3 + cc]; if (v != 0.0) { Ai[k] = foot * 5 + r; Ax[k] = v; ++k; } }
  }
  Ap[nr] = k;
  for (int f = 0; f < nfeet; ++f) for (int r = 0; r < 5; ++r) { l[5 * f + r] = s->l[5 * feet[f] + r]; u[5 * f + r] = s->u[5 * feet[f] + r]; }
  OSQPSettings settings;
  osqp_set_default_settings(&settings);
  settings.verbose = 0; settings.warm_start = 1; settings.polish = 1; settings.adaptive_rho_interval = 25;
  settings.eps_abs = 1e-3; settings.eps_rel = 1e-3;
  csc Pm = {nnzP, nr, nr, Pp, Pi, Px, -1};
  csc Am = {k, mr, nr, Ap, Ai, Ax, -1};
  OSQPData data;
  data.n = nr; data.m = mr; data.P = &Pm; data.A = &Am; data.q = q; data.l = l; data.u = u;
  OSQPWorkspace *w = 0;
  int st = -100;
  if (osqp_setup(&w, &data, &settings) == 0) {
    osqp_solve(w);
    st = (int)w->info->status_val;
    if (info) { info[0] = w->info->iter; info[1] = st; info[2] = w->info->status_polish; info[3] = w->info->rho_updates; }
    if (st == OSQP_SOLVED) for (int c = 0; c < nr; ++c) forces_out[vmap[c]] = -w->solution->x[c];
    osqp_cleanup(w);
  }
  free(vmap); free(feet); free(Pp); free(Pi); free(Px); free(Ap); free(Ai); free(Ax); free(q); free(l); free(u);
  (void)m;
  return st;
}

/* Test access: the assembled QP of the last call (dense P, q, l, u, the 5x3 cone block). */
void mpcref_get_qp(void *hd, double *P, double *q, double *l, double *u, double *cone) {
  MpcRef *s = (MpcRef *)hd;
  if (P) memcpy(P, s->P, sizeof(double) * s->n * s->n);
  if (q) memcpy(q, s->q, sizeof(double) * s->n);
  if (l) memcpy(l, s->l, sizeof(double) * s->m);
  if (u) memcpy(u, s->u, sizeof(double) * s->m);
  if (cone) memcpy(cone, s->cone, sizeof(double) * 15);
}

/* Test access: OSQP's internal (SCALED) iterates and scaling after the last solve. */
void mpcref_get_state(void *hd, double *x, double *z, double *y, double *D, double *E, double *rho_c /*[2]*/) {
  MpcRef *s = (MpcRef *)hd;
  if (!s->work) return;
  if (x) memcpy(x, s->work->x, sizeof(double) * s->n);
  if (z) memcpy(z, s->work->z, sizeof(double) * s->m);
  if (y) memcpy(y, s->work->y, sizeof(double) * s->m);
  if (D) memcpy(D, s->work->scaling->D, sizeof(double) * s->n);
  if (E) memcpy(E, s->work->scaling->E, sizeof(double) * s->m);
  if (rho_c) { rho_c[0] = s->work->settings->rho; rho_c[1] = s->work->scaling->c; }
}

/* Assembly only (no solve): for assembly parity tests. */
void mpcref_assemble_only(void *hd, const double *in) {
  MpcRef *s = (MpcRef *)hd;
  mpc_assemble(&s->mdl, in, &s->wk, s->P, s->q, s->cone, s->l, s->u);
}
void mpcref_get_dyn(void *hd, double *a_exp, double *b_exp, double *x0, double *xref) {
  MpcRef *s = (MpcRef *)hd;
  if (a_exp) memcpy(a_exp, s->wk.a_exp, sizeof(double) * 169);
  if (b_exp) memcpy(b_exp, s->wk.b_exp, sizeof(double) * 156);
  if (x0) memcpy(x0, s->wk.x0, sizeof(double) * 13);
  if (xref) memcpy(xref, s->wk.xref, sizeof(double) * 13 * s->mdl.h);
}

/* ---- batch driver (CPU baseline): one solver object per robot, static partition over threads ---- */
typedef struct {
  void **handles;
  const double *in;
  double *forces;
  int64_t *info;
  int inlen, nout, lo, hi;
} BatchJob;

static void *batch_worker(void *arg) {
  BatchJob *j = (BatchJob *)arg;
  for (int r = j->lo; r < j->hi; ++r) {
    int64_t inf[8];
    double *f = j->forces + (size_t)r * j->nout;
    int ok = mpcref_solve(j->handles[r], j->in + (size_t)r * j->inlen, f, inf, 0);
    if (!ok) for (int k = 0; k < j->nout; ++k) f[k] = NAN;
    if (j->info) memcpy(j->info + (size_t)r * 8, inf, sizeof inf);
  }
  return 0;
}

/* in: N x (56+4h) doubles; forces: N x 12h; info: N x 8 (may be NULL). */
void mpcref_batch_solve(void **handles, int N, int h, const double *in, double *forces, int64_t *info, int nthreads) {
  if (nthreads < 1) nthreads = 1;
  if (nthreads > 256) nthreads = 256;
  pthread_t th[256];
  BatchJob jobs[256];
  int per = (N + nthreads - 1) / nthreads;
  for (int t = 0; t < nthreads; ++t) {
    jobs[t] = (BatchJob){handles, in, forces, info, mpc_in_len(h), 12 * h, t * per, (t + 1) * per < N ? (t + 1) * per : N};
    if (nthreads == 1) batch_worker(&jobs[t]);
    else pthread_create(&th[t], 0, batch_worker, &jobs[t]);
  }
  if (nthreads > 1)
    for (int t = 0; t < nthreads; ++t) pthread_join(th[t], 0);
}
