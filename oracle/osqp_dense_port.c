/*
 * oracle/osqp_dense_port.c -- TEST INFRASTRUCTURE ONLY (see convex_mpc_assembly.h header).
 *
 * Scalar, dense-algebra restatement ("port") of the algorithm the reference runs for one
 * compute_contact_forces() call: mpc_osqp.cc's QP assembly (convex_mpc_assembly.h) followed by the
 * OSQP 0.6.0 solve the reference's OSQP branch performs on it.  It restates, citing the vendored
 * sources under /root/reference/extern/osqp:
 *   scaling.c:44-156        Ruiz equilibration (10 passes) + cost scaling, from scratch on every call
 *   osqp.c:1163-1265        osqp_update_P_A: unscale -> replace -> scale (NB: with the PREVIOUS q,
 *                           because the reference updates q afterwards, mpc_osqp.cc:770-777)
 *   osqp.c:751-832          osqp_update_lin_cost / osqp_update_bounds (+ auxil.c:98-141 update_rho_vec)
 *   auxil.c:79-96           set_rho_vec (equality rows get 1e3 * rho)
 *   auxil.c:164-228         ADMM iteration (rhs, x~/z~, relaxation alpha, projection, dual update)
 *   auxil.c:243-362,684-793 residuals and termination (unscaled, eps_abs = eps_rel = 1e-3)
 *   auxil.c:13-77           rho estimate / adapt_rho (adopt when >5x change; every 25 iterations)
 *   auxil.c:365-526         primal / dual infeasibility tests
 *   polish.c:19-350         polish: active-set guess from (z, y), equality-constrained re-solve,
 *                           normal-cone projection, acceptance test
 * OSQP's sparse quasi-definite KKT solve (QDLDL) is replaced by the algebraically identical reduced
 * system  (P + sigma I + A^T R A) x~ = sigma x - q + A^T (R z - y),  z~ = A x~   (dense Cholesky);
 * polish's delta-regularised KKT solve + 3 refinement steps (polish.c:102-160,258-262) by the exact
 * solve it converges to: a null-space method over the active rows (A is block diagonal, one 5x3 block
 * per (step, leg)), with minimum-norm multipliers on rank-deficient blocks (what the -delta I
 * regularisation selects).
 *
 * This file is the CPU model of the HIP kernel (same algorithm, scalar).  It is validated against the
 * real library (oracle/_ref/libconvex_mpc_ref.so) by tests/test_oracle.py::test_dense_port_tracks_vendored_osqp.  Built twice:
 * -DREAL=double (checker) and -DREAL=float (predicts the effect of fp32 device arithmetic).
 */
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#ifndef REAL
#define REAL double
#endif
#define AREAL REAL
#include "convex_mpc_assembly.h"

/* include/constants.h:59-88 */
#define Q_RHO 0.1
#define Q_SIGMA 1e-6
#define Q_MAX_ITER 4000
#define Q_EPS_ABS 1e-3 /* mpc_osqp.cc:711 */
#define Q_EPS_REL 1e-3 /* mpc_osqp.cc:712 */
#define Q_EPS_PRIM_INF 1e-4
#define Q_EPS_DUAL_INF 1e-4
#define Q_ALPHA 1.6
#define Q_RHO_MIN 1e-6
#define Q_RHO_MAX 1e6
#define Q_RHO_EQ_OVER_INEQ 1e3
#define Q_RHO_TOL 1e-4
#define Q_CHECK 25          /* CHECK_TERMINATION and adaptive_rho_interval (mpc_osqp.cc:710) */
#define Q_SCALING_ITERS 10
#define Q_MIN_SCALING 1e-4
#define Q_MAX_SCALING 1e4
#define Q_ADAPT_TOL 5.0
#define Q_INFTY 1e30

enum { ST_SOLVED = 1, ST_SOLVED_INACCURATE = 2, ST_PRIMAL_INF_INACC = 3, ST_DUAL_INF_INACC = 4, ST_MAX_ITER = -2,
       ST_PRIMAL_INF = -3, ST_DUAL_INF = -4, ST_NON_CVX = -7, ST_UNSOLVED = -10 };

typedef struct {
  int h, n, m, nf;
  MpcModel mdl;
  MpcWork wk;
  REAL *P, *q, *l, *u, cone[15];      /* unscaled problem of this call */
  REAL *q_old;                        /* unscaled q of the previous call (scale_data quirk) */
  REAL *Ps, *qs, *ls, *us, *As;       /* scaled problem; As = nf blocks of 5x3 */
  REAL *D, *Dinv, *E, *Einv, c, cinv;
  REAL *x, *z, *y, *xp, *zp, *xt, *zt, *dx, *dy;
  REAL *Ax, *Px, *Aty, *tn, *tm;
  REAL rho, *rho_vec, *rho_inv;
  int *ctype;
  REAL *K;                            /* Cholesky factor (lower) of P + sigma I + A^T R A */
  REAL *W;                            /* polish scratch */
  int first_run;
  int iter, status, status_polish, rho_updates, nfact;
  REAL pri_res, dua_res;
} Port;

static REAL *ralloc(size_t k) { return (REAL *)calloc(k ? k : 1, sizeof(REAL)); }

void *port_create(double mass, const double *inertia9, int h, double dt, double alpha) {
  if (h < 2 || h > MPC_MAX_H) return 0;
  Port *s = (Port *)calloc(1, sizeof(Port));
  s->h = h; s->n = 12 * h; s->m = 20 * h; s->nf = 4 * h;
  mpc_model_init(&s->mdl, mass, inertia9, h, dt, alpha);
  const int n = s->n, m = s->m;
  s->P = ralloc((size_t)n * n); s->q = ralloc(n); s->l = ralloc(m); s->u = ralloc(m); s->q_old = ralloc(n);
  s->Ps = ralloc((size_t)n * n); s->qs = ralloc(n); s->ls = ralloc(m); s->us = ralloc(m); s->As = ralloc(15 * s->nf);
  s->D = ralloc(n); s->Dinv = ralloc(n); s->E = ralloc(m); s->Einv = ralloc(m);
  s->x = ralloc(n); s->z = ralloc(m); s->y = ralloc(m); s->xp = ralloc(n); s->zp = ralloc(m);
  s->xt = ralloc(n); s->zt = ralloc(m); s->dx = ralloc(n); s->dy = ralloc(m);
  s->Ax = ralloc(m); s->Px = ralloc(n); s->Aty = ralloc(n); s->tn = ralloc(n); s->tm = ralloc(m);
  s->rho_vec = ralloc(m); s->rho_inv = ralloc(m); s->ctype = (int *)calloc(m, sizeof(int));
  s->K = ralloc((size_t)n * n);
  s->W = ralloc((size_t)n * n + 64 * n);
  s->rho = (REAL)Q_RHO;
  s->first_run = 1;
  s->status = ST_UNSOLVED;
  return s;
}

void port_destroy(void *hd) {
  Port *s = (Port *)hd;
  if (!s) return;
  REAL *ptrs[] = {s->P, s->q, s->l, s->u, s->q_old, s->Ps, s->qs, s->ls, s->us, s->As, s->D, s->Dinv, s->E, s->Einv,
                  s->x, s->z, s->y, s->xp, s->zp, s->xt, s->zt, s->dx, s->dy, s->Ax, s->Px, s->Aty, s->tn, s->tm,
                  s->rho_vec, s->rho_inv, s->K, s->W};
  for (size_t i = 0; i < sizeof ptrs / sizeof *ptrs; ++i) free(ptrs[i]);
  free(s->ctype);
  free(s);
}

/* ---- small dense helpers on the block structure ------------------------------------------------ */
static REAL norm_inf(const REAL *v, int k) {
  REAL mx = 0;
  for (int i = 0; i < k; ++i) { REAL a = fabs(v[i]); if (a > mx) mx = a; }
  return mx;
}
static REAL scaled_norm_inf(const REAL *s, const REAL *v, int k) { /* lin_alg.c vec_scaled_norm_inf */
  REAL mx = 0;
  for (int i = 0; i < k; ++i) { REAL a = fabs(s[i] * v[i]); if (a > mx) mx = a; }
  return mx;
}
/* out(m) = A x  with the scaled block-diagonal A */
static void mul_A(const Port *s, const REAL *x, REAL *out) {
  for (int f = 0; f < s->nf; ++f) {
    const REAL *a = s->As + 15 * f, *xf = x + 3 * f;
    for (int r = 0; r < 5; ++r) out[5 * f + r] = a[r * 3] * xf[0] + a[r * 3 + 1] * xf[1] + a[r * 3 + 2] * xf[2];
  }
}
/* out(n) = A^T y */
static void mul_At(const Port *s, const REAL *y, REAL *out) {
  for (int f = 0; f < s->nf; ++f) {
    const REAL *a = s->As + 15 * f, *yf = y + 5 * f;
    for (int c = 0; c < 3; ++c) {
      REAL t = 0;
      for (int r = 0; r < 5; ++r) t += a[r * 3 + c] * yf[r];
      out[3 * f + c] = t;
    }
  }
}
static void mul_P(const Port *s, const REAL *x, REAL *out) {
  const int n = s->n;
  for (int i = 0; i < n; ++i) {
    REAL t = 0;
    const REAL *row = s->Ps + (size_t)i * n;
    for (int j = 0; j < n; ++j) t += row[j] * x[j];
    out[i] = t;
  }
}
static REAL limit_scaling(REAL v) { /* scaling.c:7-14 */
  v = v < (REAL)Q_MIN_SCALING ? (REAL)1.0 : v;
  v = v > (REAL)Q_MAX_SCALING ? (REAL)Q_MAX_SCALING : v;
  return v;
}

/* scaling.c:44-156 scale_data on (Ps, As, qs); then l, u.  qs must hold the q to equilibrate with. */
static void scale_data(Port *s) {
  const int n = s->n, m = s->m, nf = s->nf;
  s->c = 1;
  for (int i = 0; i < n; ++i) s->D[i] = 1;
  for (int i = 0; i < m; ++i) s->E[i] = 1;
  REAL *dt = s->tn, *et = s->tm;
  for (int it = 0; it < Q_SCALING_ITERS; ++it) {
    /* scaling.c:28-42 inf-norm of the KKT columns */
    for (int j = 0; j < n; ++j) {
      REAL mx = 0;
      for (int i = 0; i < n; ++i) { REAL a = fabs(s->Ps[(size_t)i * n + j]); if (a > mx) mx = a; }
      const REAL *a = s->As + 15 * (j / 3);
      for (int r = 0; r < 5; ++r) { REAL v = fabs(a[r * 3 + j % 3]); if (v > mx) mx = v; }
      dt[j] = mx;
    }
    for (int f = 0; f < nf; ++f)
      for (int r = 0; r < 5; ++r) {
        const REAL *a = s->As + 15 * f + 3 * r;
        REAL mx = fabs(a[0]);
        if (fabs(a[1]) > mx) mx = fabs(a[1]);
        if (fabs(a[2]) > mx) mx = fabs(a[2]);
        et[5 * f + r] = mx;
      }
    for (int j = 0; j < n; ++j) dt[j] = (REAL)1.0 / sqrt(limit_scaling(dt[j]));
    for (int i = 0; i < m; ++i) et[i] = (REAL)1.0 / sqrt(limit_scaling(et[i]));
    /* P <- D P D, A <- E A D, q <- D q */
    for (int i = 0; i < n; ++i)
      for (int j = 0; j < n; ++j) s->Ps[(size_t)i * n + j] *= dt[i] * dt[j];
    for (int f = 0; f < nf; ++f)
      for (int r = 0; r < 5; ++r)
        for (int c = 0; c < 3; ++c) s->As[15 * f + 3 * r + c] *= et[5 * f + r] * dt[3 * f + c];
    for (int j = 0; j < n; ++j) { s->qs[j] *= dt[j]; s->D[j] *= dt[j]; }
    for (int i = 0; i < m; ++i) s->E[i] *= et[i];
    /* cost normalisation, scaling.c:108-139 */
    REAL mean = 0;
    for (int j = 0; j < n; ++j) {
      REAL mx = 0;
      for (int i = 0; i < n; ++i) { REAL a = fabs(s->Ps[(size_t)i * n + j]); if (a > mx) mx = a; }
      mean += mx;
    }
    mean /= n;
    REAL nq = limit_scaling(norm_inf(s->qs, n));
    REAL ct = mean > nq ? mean : nq;
    ct = (REAL)1.0 / limit_scaling(ct);
    for (size_t k = 0; k < (size_t)n * n; ++k) s->Ps[k] *= ct;
    for (int j = 0; j < n; ++j) s->qs[j] *= ct;
    s->c *= ct;
  }
  s->cinv = (REAL)1.0 / s->c;
  for (int j = 0; j < n; ++j) s->Dinv[j] = (REAL)1.0 / s->D[j];
  for (int i = 0; i < m; ++i) s->Einv[i] = (REAL)1.0 / s->E[i];
}

/* Cholesky of K = Ps + sigma I + A^T diag(rho_vec) A (lower triangle, in s->K).  Returns 0 if SPD. */
static int factor_K(Port *s) {
  const int n = s->n;
  REAL *K = s->K;
  memcpy(K, s->Ps, sizeof(REAL) * n * n);
  for (int i = 0; i < n; ++i) K[(size_t)i * n + i] += (REAL)Q_SIGMA;
  for (int f = 0; f < s->nf; ++f) {
    const REAL *a = s->As + 15 * f;
    for (int c1 = 0; c1 < 3; ++c1)
      for (int c2 = 0; c2 < 3; ++c2) {
        REAL t = 0;
        for (int r = 0; r < 5; ++r) t += a[r * 3 + c1] * s->rho_vec[5 * f + r] * a[r * 3 + c2];
        K[(size_t)(3 * f + c1) * n + 3 * f + c2] += t;
      }
  }
  for (int j = 0; j < n; ++j) {
    REAL d = K[(size_t)j * n + j];
    for (int k = 0; k < j; ++k) d -= K[(size_t)j * n + k] * K[(size_t)j * n + k];
    if (!(d > 0)) return 1;
    d = sqrt(d);
    K[(size_t)j * n + j] = d;
    for (int i = j + 1; i < n; ++i) {
      REAL t = K[(size_t)i * n + j];
      for (int k = 0; k < j; ++k) t -= K[(size_t)i * n + k] * K[(size_t)j * n + k];
      K[(size_t)i * n + j] = t / d;
    }
  }
  s->nfact++;
  return 0;
}
static void chol_solve(const REAL *L, int n, int ld, REAL *b) {
  for (int i = 0; i < n; ++i) {
    REAL t = b[i];
    for (int k = 0; k < i; ++k) t -= L[(size_t)i * ld + k] * b[k];
    b[i] = t / L[(size_t)i * ld + i];
  }
  for (int i = n - 1; i >= 0; --i) {
    REAL t = b[i];
    for (int k = i + 1; k < n; ++k) t -= L[(size_t)k * ld + i] * b[k];
    b[i] = t / L[(size_t)i * ld + i];
  }
}

/* auxil.c:79-96 set_rho_vec (first run) / auxil.c:98-141 update_rho_vec (later runs).  Returns
 * whether any constraint type changed. */
static int classify_rows(Port *s, int first) {
  int changed = 0;
  if (first) {
    s->rho = s->rho < (REAL)Q_RHO_MIN ? (REAL)Q_RHO_MIN : (s->rho > (REAL)Q_RHO_MAX ? (REAL)Q_RHO_MAX : s->rho);
  }
  for (int i = 0; i < s->m; ++i) {
    int t;
    if (s->ls[i] < -(REAL)(Q_INFTY * Q_MIN_SCALING) && s->us[i] > (REAL)(Q_INFTY * Q_MIN_SCALING)) t = -1;
    else if (s->us[i] - s->ls[i] < (REAL)Q_RHO_TOL) t = 1;
    else t = 0;
    if (first || t != s->ctype[i]) {
      s->ctype[i] = t;
      s->rho_vec[i] = t == -1 ? (REAL)Q_RHO_MIN : (t == 1 ? (REAL)Q_RHO_EQ_OVER_INEQ * s->rho : s->rho);
      s->rho_inv[i] = (REAL)1.0 / s->rho_vec[i];
      changed = 1;
    }
  }
  return changed;
}

/* osqp.c:1267-1310 osqp_update_rho */
static int update_rho(Port *s, REAL rho_new) {
  s->rho = rho_new < (REAL)Q_RHO_MIN ? (REAL)Q_RHO_MIN : (rho_new > (REAL)Q_RHO_MAX ? (REAL)Q_RHO_MAX : rho_new);
  for (int i = 0; i < s->m; ++i) {
    if (s->ctype[i] == 0) { s->rho_vec[i] = s->rho; s->rho_inv[i] = (REAL)1.0 / s->rho; }
    else if (s->ctype[i] == 1) { s->rho_vec[i] = (REAL)Q_RHO_EQ_OVER_INEQ * s->rho; s->rho_inv[i] = (REAL)1.0 / s->rho_vec[i]; }
  }
  return factor_K(s);
}

/* auxil.c:563-629 update_info for the ADMM iterates: residuals (unscaled).  Side effects kept:
 * zp <- A x - z and xp <- q + P x + A^T y (compute_pri_res / compute_dua_res use them as scratch and
 * compute_rho_estimate reads them, auxil.c:23-24). */
static void update_info(Port *s, const REAL *x, const REAL *z, const REAL *y, REAL *pri, REAL *dua) {
  const int n = s->n, m = s->m;
  mul_A(s, x, s->Ax);
  for (int i = 0; i < m; ++i) s->zp[i] = s->Ax[i] - z[i];
  *pri = scaled_norm_inf(s->Einv, s->zp, m);
  mul_P(s, x, s->Px);
  mul_At(s, y, s->Aty);
  for (int i = 0; i < n; ++i) s->xp[i] = s->qs[i] + s->Px[i] + s->Aty[i];
  *dua = s->cinv * scaled_norm_inf(s->Dinv, s->xp, n);
}

/* auxil.c:365-428 */
static int is_primal_infeasible(Port *s, REAL eps) {
  const int m = s->m, n = s->n;
  for (int i = 0; i < m; ++i) {
    if (s->us[i] > (REAL)(Q_INFTY * Q_MIN_SCALING)) {
      if (s->ls[i] < -(REAL)(Q_INFTY * Q_MIN_SCALING)) s->dy[i] = 0;
      else s->dy[i] = s->dy[i] < 0 ? s->dy[i] : 0;
    } else if (s->ls[i] < -(REAL)(Q_INFTY * Q_MIN_SCALING)) s->dy[i] = s->dy[i] > 0 ? s->dy[i] : 0;
  }
  REAL nd = scaled_norm_inf(s->E, s->dy, m);
  if (nd > eps) {
    REAL lhs = 0;
    for (int i = 0; i < m; ++i) lhs += s->us[i] * (s->dy[i] > 0 ? s->dy[i] : 0) + s->ls[i] * (s->dy[i] < 0 ? s->dy[i] : 0);
    if (lhs < -eps * nd) {
      mul_At(s, s->dy, s->tn);
      return scaled_norm_inf(s->Dinv, s->tn, n) < eps * nd;
    }
  }
  return 0;
}
/* auxil.c:430-526 */
static int is_dual_infeasible(Port *s, REAL eps) {
  const int m = s->m, n = s->n;
  REAL nd = scaled_norm_inf(s->D, s->dx, n);
  if (nd > eps) {
    REAL qd = 0;
    for (int i = 0; i < n; ++i) qd += s->qs[i] * s->dx[i];
    if (qd < -s->c * eps * nd) {
      mul_P(s, s->dx, s->tn);
      if (scaled_norm_inf(s->Dinv, s->tn, n) < s->c * eps * nd) {
        mul_A(s, s->dx, s->tm);
        for (int i = 0; i < m; ++i) {
          REAL v = s->Einv[i] * s->tm[i];
          if ((s->us[i] < (REAL)(Q_INFTY * Q_MIN_SCALING) && v > eps * nd) ||
              (s->ls[i] > -(REAL)(Q_INFTY * Q_MIN_SCALING) && v < -eps * nd)) return 0;
        }
        return 1;
      }
    }
  }
  return 0;
}

/* auxil.c:684-793 check_termination */
static int check_termination(Port *s, int approximate) {
  REAL ea = (REAL)Q_EPS_ABS, er = (REAL)Q_EPS_REL, epi = (REAL)Q_EPS_PRIM_INF, edi = (REAL)Q_EPS_DUAL_INF;
  if (s->pri_res > (REAL)Q_INFTY || s->dua_res > (REAL)Q_INFTY) { s->status = ST_NON_CVX; return 1; }
  if (approximate) { ea *= 10; er *= 10; epi *= 10; edi *= 10; }
  int prim_ok = 0, dual_ok = 0, prim_inf = 0, dual_inf = 0;
  REAL a = scaled_norm_inf(s->Einv, s->z, s->m), b = scaled_norm_inf(s->Einv, s->Ax, s->m);
  REAL eps_prim = ea + er * (a > b ? a : b);
  if (s->pri_res < eps_prim) prim_ok = 1; else prim_inf = is_primal_infeasible(s, epi);
  REAL d0 = scaled_norm_inf(s->Dinv, s->qs, s->n), d1 = scaled_norm_inf(s->Dinv, s->Aty, s->n),
       d2 = scaled_norm_inf(s->Dinv, s->Px, s->n);
  REAL dm = d0 > d1 ? d0 : d1;
  dm = dm > d2 ? dm : d2;
  REAL eps_dual = ea + er * s->cinv * dm;
  if (s->dua_res < eps_dual) dual_ok = 1; else dual_inf = is_dual_infeasible(s, edi);
  if (prim_ok && dual_ok) { s->status = approximate ? ST_SOLVED_INACCURATE : ST_SOLVED; return 1; }
  if (prim_inf) { s->status = approximate ? ST_PRIMAL_INF_INACC : ST_PRIMAL_INF; return 1; }
  if (dual_inf) { s->status = approximate ? ST_DUAL_INF_INACC : ST_DUAL_INF; return 1; }
  return 0;
}

/* auxil.c:13-55 compute_rho_estimate (reads zp / xp left behind by update_info) */
static REAL rho_estimate(Port *s) {
  REAL pri = norm_inf(s->zp, s->m), dua = norm_inf(s->xp, s->n);
  REAL a = norm_inf(s->z, s->m), b = norm_inf(s->Ax, s->m);
  pri /= ((a > b ? a : b) + (REAL)1e-10);
  REAL d0 = norm_inf(s->qs, s->n), d1 = norm_inf(s->Aty, s->n), d2 = norm_inf(s->Px, s->n);
  REAL dm = d0 > d1 ? d0 : d1;
  dm = dm > d2 ? dm : d2;
  dua /= (dm + (REAL)1e-10);
  REAL r = s->rho * sqrt(pri / (dua + (REAL)1e-10));
  return r < (REAL)Q_RHO_MIN ? (REAL)Q_RHO_MIN : (r > (REAL)Q_RHO_MAX ? (REAL)Q_RHO_MAX : r);
}

/* ---- polish (polish.c:232-350), null-space form -------------------------------------------------
 * Active rows: lower if z - l < -y, upper if u - z < y (polish.c:36-52, scaled iterates).
 * Per 5x3 block: modified Gram-Schmidt over the active rows gives an orthonormal basis Q (rank r<=3)
 * of their span, the minimum-norm point x0 satisfying the independent active rows, and the
 * orthogonal complement N (3 x (3-r)).  Reduced SPD system (N^T P N) w = -N^T (q + P x0).
 */
static int polish(Port *s) {
  const int n = s->n, m = s->m, nf = s->nf;
  REAL *Nb = s->W;                 /* n x 3 : per foot up to 3 null vectors, each of length 3 (stored 9 per foot) */
  REAL *Qb = Nb + 9 * nf;          /* per foot up to 3 range vectors (9 per foot) */
  REAL *x0 = Qb + 9 * nf;          /* n */
  REAL *xpol = x0 + n, *ypol = xpol + n, *zpol = ypol + m, *g = zpol + m, *rhs = g + n; /* rhs: n */
  REAL *H = rhs + n;               /* nfree x nfree */
  int *rk = (int *)calloc(nf, sizeof(int)), *act = (int *)calloc(m, sizeof(int)), *col0 = (int *)calloc(nf + 1, sizeof(int));
  int nfree = 0;
  for (int i = 0; i < m; ++i) {
    if (s->z[i] - s->ls[i] < -s->y[i]) act[i] = -1;         /* lower-active */
    else if (s->us[i] - s->z[i] < s->y[i]) act[i] = 1;      /* upper-active */
    else act[i] = 0;
  }
  for (int f = 0; f < nf; ++f) {
    const REAL *a = s->As + 15 * f;
    REAL *Q = Qb + 9 * f, *N = Nb + 9 * f, cq[3] = {0, 0, 0};
    int r = 0;
    for (int row = 0; row < 5 && r < 3; ++row) {
      if (!act[5 * f + row]) continue;
      REAL v[3] = {a[row * 3], a[row * 3 + 1], a[row * 3 + 2]};
      REAL nrm0 = sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
      REAL tgt = act[5 * f + row] < 0 ? s->ls[5 * f + row] : s->us[5 * f + row];
      for (int pass = 0; pass < 2; ++pass)  /* re-orthogonalise once */
        for (int k = 0; k < r; ++k) {
          REAL d = v[0] * Q[3 * k] + v[1] * Q[3 * k + 1] + v[2] * Q[3 * k + 2];
          v[0] -= d * Q[3 * k]; v[1] -= d * Q[3 * k + 1]; v[2] -= d * Q[3 * k + 2];
        }
      REAL nrm = sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
      if (!(nrm > (REAL)1e-6 * nrm0)) continue;             /* dependent row */
      /* a_row . x0 so far */
      REAL ax = 0;
      for (int k = 0; k < r; ++k)
        ax += cq[k] * (a[row * 3] * Q[3 * k] + a[row * 3 + 1] * Q[3 * k + 1] + a[row * 3 + 2] * Q[3 * k + 2]);
      Q[3 * r] = v[0] / nrm; Q[3 * r + 1] = v[1] / nrm; Q[3 * r + 2] = v[2] / nrm;
      REAL aq = a[row * 3] * Q[3 * r] + a[row * 3 + 1] * Q[3 * r + 1] + a[row * 3 + 2] * Q[3 * r + 2];
      cq[r] = (tgt - ax) / aq;
      ++r;
    }
    rk[f] = r;
    for (int c = 0; c < 3; ++c) {
      REAL t = 0;
      for (int k = 0; k < r; ++k) t += cq[k] * Q[3 * k + c];
      x0[3 * f + c] = t;
    }
    /* complement basis */
    int nn = 0;
    if (r == 0) { REAL I3[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}; memcpy(N, I3, sizeof I3); nn = 3; }
    else if (r == 1) {
      /* two vectors orthogonal to Q0 */
      const REAL *q0 = Q;
      int imin = fabs(q0[0]) <= fabs(q0[1]) ? (fabs(q0[0]) <= fabs(q0[2]) ? 0 : 2) : (fabs(q0[1]) <= fabs(q0[2]) ? 1 : 2);
      REAL e[3] = {0, 0, 0};
      e[imin] = 1;
      REAL d = q0[imin];
      REAL v[3] = {e[0] - d * q0[0], e[1] - d * q0[1], e[2] - d * q0[2]};
      REAL nr = sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
      N[0] = v[0] / nr; N[1] = v[1] / nr; N[2] = v[2] / nr;
      N[3] = q0[1] * N[2] - q0[2] * N[1]; N[4] = q0[2] * N[0] - q0[0] * N[2]; N[5] = q0[0] * N[1] - q0[1] * N[0];
      nn = 2;
    } else if (r == 2) {
      N[0] = Q[1] * Q[5] - Q[2] * Q[4]; N[1] = Q[2] * Q[3] - Q[0] * Q[5]; N[2] = Q[0] * Q[4] - Q[1] * Q[3];
      REAL nr = sqrt(N[0] * N[0] + N[1] * N[1] + N[2] * N[2]);
      N[0] /= nr; N[1] /= nr; N[2] /= nr;
      nn = 1;
    }
    col0[f] = nfree;
    nfree += nn;
  }
  col0[nf] = nfree;
  /* g0 = q + P x0 */
  mul_P(s, x0, g);
  for (int i = 0; i < n; ++i) g[i] += s->qs[i];
  /* H = N^T P N (nfree x nfree), rhs = -N^T g0 */
  for (int f1 = 0; f1 < nf; ++f1)
    for (int k1 = 0; k1 < col0[f1 + 1] - col0[f1]; ++k1) {
      const REAL *n1 = Nb + 9 * f1 + 3 * k1;
      const int i1 = col0[f1] + k1;
      /* t = P[:, foot f1 cols] n1  -> vector of length n */
      for (int i = 0; i < n; ++i) {
        const REAL *row = s->Ps + (size_t)i * n + 3 * f1;
        s->tn[i] = row[0] * n1[0] + row[1] * n1[1] + row[2] * n1[2];
      }
      for (int f2 = 0; f2 < nf; ++f2)
        for (int k2 = 0; k2 < col0[f2 + 1] - col0[f2]; ++k2) {
          const REAL *n2 = Nb + 9 * f2 + 3 * k2;
          H[(size_t)(col0[f2] + k2) * nfree + i1] = n2[0] * s->tn[3 * f2] + n2[1] * s->tn[3 * f2 + 1] + n2[2] * s->tn[3 * f2 + 2];
        }
      rhs[i1] = -(n1[0] * g[3 * f1] + n1[1] * g[3 * f1 + 1] + n1[2] * g[3 * f1 + 2]);
    }
  /* Cholesky of H in place */
  int bad = 0;
  for (int j = 0; j < nfree && !bad; ++j) {
    REAL d = H[(size_t)j * nfree + j];
    for (int k = 0; k < j; ++k) d -= H[(size_t)j * nfree + k] * H[(size_t)j * nfree + k];
    if (!(d > 0)) { bad = 1; break; }
    d = sqrt(d);
    H[(size_t)j * nfree + j] = d;
    for (int i = j + 1; i < nfree; ++i) {
      REAL t = H[(size_t)i * nfree + j];
      for (int k = 0; k < j; ++k) t -= H[(size_t)i * nfree + k] * H[(size_t)j * nfree + k];
      H[(size_t)i * nfree + j] = t / d;
    }
  }
  int ok = 0;
  if (!bad) {
    s->nfact++;
    chol_solve(H, nfree, nfree, rhs);
    for (int f = 0; f < nf; ++f)
      for (int c = 0; c < 3; ++c) {
        REAL t = x0[3 * f + c];
        for (int k = 0; k < col0[f + 1] - col0[f]; ++k) t += Nb[9 * f + 3 * k + c] * rhs[col0[f] + k];
        xpol[3 * f + c] = t;
      }
    /* multipliers: A_act^T y_act = -(P x + q) per block, minimum norm (polish.c:162-190 get_ypol) */
    mul_P(s, xpol, g);
    for (int i = 0; i < n; ++i) g[i] = -(g[i] + s->qs[i]);
    for (int f = 0; f < nf; ++f) {
      const REAL *a = s->As + 15 * f, *Q = Qb + 9 * f;
      const int r = rk[f];
      REAL AQ[15], M[9], gam[3], sol[3];
      for (int row = 0; row < 5; ++row) ypol[5 * f + row] = 0;
      if (r == 0) continue;
      for (int row = 0; row < 5; ++row)
        for (int k = 0; k < r; ++k)
          AQ[row * 3 + k] = act[5 * f + row] ? a[row * 3] * Q[3 * k] + a[row * 3 + 1] * Q[3 * k + 1] + a[row * 3 + 2] * Q[3 * k + 2] : 0;
      for (int k1 = 0; k1 < r; ++k1) {
        for (int k2 = 0; k2 < r; ++k2) {
          REAL t = 0;
          for (int row = 0; row < 5; ++row) t += AQ[row * 3 + k1] * AQ[row * 3 + k2];
          M[k1 * 3 + k2] = t;
        }
        gam[k1] = Q[3 * k1] * g[3 * f] + Q[3 * k1 + 1] * g[3 * f + 1] + Q[3 * k1 + 2] * g[3 * f + 2];
      }
      /* solve M sol = gam (r x r SPD) by Gaussian elimination */
      REAL Mm[9];
      memcpy(Mm, M, sizeof Mm);
      for (int k = 0; k < r; ++k) sol[k] = gam[k];
      for (int p = 0; p < r; ++p) {
        for (int i = p + 1; i < r; ++i) {
          REAL fct = Mm[i * 3 + p] / Mm[p * 3 + p];
          for (int j2 = p; j2 < r; ++j2) Mm[i * 3 + j2] -= fct * Mm[p * 3 + j2];
          sol[i] -= fct * sol[p];
        }
      }
      for (int p = r - 1; p >= 0; --p) {
        REAL t = sol[p];
        for (int j2 = p + 1; j2 < r; ++j2) t -= Mm[p * 3 + j2] * sol[j2];
        sol[p] = t / Mm[p * 3 + p];
      }
      for (int row = 0; row < 5; ++row) {
        REAL t = 0;
        for (int k = 0; k < r; ++k) t += AQ[row * 3 + k] * sol[k];
        ypol[5 * f + row] = t;
      }
    }
    /* polish.c:300-304: z = A x; project_normalcone (proj.c:17-31) */
    mul_A(s, xpol, zpol);
    for (int i = 0; i < m; ++i) {
      REAL t = zpol[i] + ypol[i];
      REAL zc = t < s->ls[i] ? s->ls[i] : (t > s->us[i] ? s->us[i] : t);
      zpol[i] = zc;
      ypol[i] = t - zc;
    }
    /* update_info(polish=1): residuals at the polished point (uses zp/xp as scratch) */
    REAL pri, dua;
    REAL savez[1]; (void)savez;
    update_info(s, xpol, zpol, ypol, &pri, &dua);
    ok = (pri < s->pri_res && dua < s->dua_res) || (pri < s->pri_res && s->dua_res < (REAL)1e-10) ||
         (dua < s->dua_res && s->pri_res < (REAL)1e-10); /* polish.c:309-322 */
    if (ok) {
      s->pri_res = pri; s->dua_res = dua;
      memcpy(s->x, xpol, sizeof(REAL) * n);
      memcpy(s->z, zpol, sizeof(REAL) * m);
      memcpy(s->y, ypol, sizeof(REAL) * m);
    }
  }
  s->status_polish = ok ? 1 : -1;
  free(rk); free(act); free(col0);
  return ok;
}


/* ---- polish, literal form (polish.c:232-350): dense LU of the delta-regularised reduced KKT
 * [[P + delta I, Ared^T], [Ared, -delta I]] + POLISH_REFINE_ITER = 3 refinement steps
 * (polish.c:102-160).  O((n+mred)^3): checker only. */
#define Q_DELTA 1e-6
#define Q_POLISH_REFINE 3
static void lu_factor(REAL *A, int *piv, int nn) {
  for (int k = 0; k < nn; ++k) {
    int p = k; REAL mx = fabs(A[(size_t)k * nn + k]);
    for (int i = k + 1; i < nn; ++i) if (fabs(A[(size_t)i * nn + k]) > mx) { mx = fabs(A[(size_t)i * nn + k]); p = i; }
    piv[k] = p;
    if (p != k) for (int j = 0; j < nn; ++j) { REAL t = A[(size_t)k * nn + j]; A[(size_t)k * nn + j] = A[(size_t)p * nn + j]; A[(size_t)p * nn + j] = t; }
    REAL d = A[(size_t)k * nn + k];
    for (int i = k + 1; i < nn; ++i) {
      REAL f = A[(size_t)i * nn + k] / d;
      A[(size_t)i * nn + k] = f;
      for (int j = k + 1; j < nn; ++j) A[(size_t)i * nn + j] -= f * A[(size_t)k * nn + j];
    }
  }
}
static void lu_solve(const REAL *A, const int *piv, int nn, REAL *b) {
  for (int k = 0; k < nn; ++k) if (piv[k] != k) { REAL t = b[k]; b[k] = b[piv[k]]; b[piv[k]] = t; }
  for (int k = 0; k < nn; ++k) for (int i = k + 1; i < nn; ++i) b[i] -= A[(size_t)i * nn + k] * b[k];
  for (int i = nn - 1; i >= 0; --i) { REAL t = b[i]; for (int j = i + 1; j < nn; ++j) t -= A[(size_t)i * nn + j] * b[j]; b[i] = t / A[(size_t)i * nn + i]; }
}
static int polish_kkt(Port *s) {
  const int n = s->n, m = s->m;
  int *rows = (int *)malloc(sizeof(int) * 2 * m), *isup = (int *)malloc(sizeof(int) * 2 * m), mred = 0;
  for (int i = 0; i < m; ++i) if (s->z[i] - s->ls[i] < -s->y[i]) { rows[mred] = i; isup[mred++] = 0; }
  for (int i = 0; i < m; ++i) if (s->us[i] - s->z[i] < s->y[i]) { int dup = 0; for (int k = 0; k < mred; ++k) if (rows[k] == i) dup = 1; if (!dup) { rows[mred] = i; isup[mred++] = 1; } }
  const int nn = n + mred;
  REAL *K = ralloc((size_t)nn * nn), *Kr = ralloc((size_t)nn * nn), *rhs = ralloc(nn), *sol = ralloc(nn), *res = ralloc(nn);
  int *piv = (int *)malloc(sizeof(int) * nn);
  for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) K[(size_t)i * nn + j] = s->Ps[(size_t)i * n + j];
  for (int k = 0; k < mred; ++k) { int r = rows[k], f = r / 5, rr = r % 5;
    for (int c = 0; c < 3; ++c) { REAL v = s->As[15 * f + 3 * rr + c]; K[(size_t)(n + k) * nn + 3 * f + c] = v; K[(size_t)(3 * f + c) * nn + n + k] = v; } }
  memcpy(Kr, K, sizeof(REAL) * nn * nn);
  for (int i = 0; i < n; ++i) Kr[(size_t)i * nn + i] += (REAL)Q_DELTA;
  for (int k = 0; k < mred; ++k) Kr[(size_t)(n + k) * nn + n + k] -= (REAL)Q_DELTA;
  lu_factor(Kr, piv, nn);
  for (int i = 0; i < n; ++i) rhs[i] = -s->qs[i];
  for (int k = 0; k < mred; ++k) rhs[n + k] = isup[k] ? s->us[rows[k]] : s->ls[rows[k]];
  memcpy(sol, rhs, sizeof(REAL) * nn);
  lu_solve(Kr, piv, nn, sol);
  for (int it = 0; it < Q_POLISH_REFINE; ++it) {
    for (int i = 0; i < nn; ++i) { REAL t = rhs[i]; for (int j = 0; j < nn; ++j) t -= K[(size_t)i * nn + j] * sol[j]; res[i] = t; }
    lu_solve(Kr, piv, nn, res);
    for (int i = 0; i < nn; ++i) sol[i] += res[i];
  }
  REAL *xpol = s->W, *ypol = xpol + n, *zpol = ypol + m;
  memcpy(xpol, sol, sizeof(REAL) * n);
  for (int i = 0; i < m; ++i) ypol[i] = 0;
  for (int k = 0; k < mred; ++k) ypol[rows[k]] = sol[n + k];
  mul_A(s, xpol, zpol);
  for (int i = 0; i < m; ++i) { REAL t = zpol[i] + ypol[i]; REAL zc = t < s->ls[i] ? s->ls[i] : (t > s->us[i] ? s->us[i] : t); zpol[i] = zc; ypol[i] = t - zc; }
  REAL pri, dua;
  update_info(s, xpol, zpol, ypol, &pri, &dua);
  int ok = (pri < s->pri_res && dua < s->dua_res) || (pri < s->pri_res && s->dua_res < (REAL)1e-10) || (dua < s->dua_res && s->pri_res < (REAL)1e-10);
  if (ok) { s->pri_res = pri; s->dua_res = dua; memcpy(s->x, xpol, sizeof(REAL) * n); memcpy(s->z, zpol, sizeof(REAL) * m); memcpy(s->y, ypol, sizeof(REAL) * m); }
  s->status_polish = ok ? 1 : -1;
  s->nfact++;
  free(rows); free(isup); free(K); free(Kr); free(rhs); free(sol); free(res); free(piv);
  return ok;
}


/* ---- polish, reduced-coordinate form of the SAME delta-regularised iteration (the form the HIP
 * kernel runs).  Per 5x3 block f with active rows A_f: Q_f = orthonormal basis of the active rows'
 * span (rank r_f), N_f its complement, Gamma_f = pinv(A_f^T A_f) (3x3).  With block-diagonal
 * Gamma, N and u = Gamma A^T r2, one application (x, y) = Kreg^{-1} (r1, r2) is, to O(delta^2):
 *   tf  = Gamma (r1 - P u - delta u)
 *   H_d = N^T (P + delta I - delta P Gamma P) N ;  w = H_d^{-1} N^T (r1 - P u - delta P tf)
 *   xN  = N w ;  x = u + delta (tf - Gamma P xN) + xN
 *   y   = A Gamma (r1 - P u - delta u - P xN)  -  (r2 - A u) / delta      (active rows only)
 * followed by POLISH_REFINE_ITER refinement steps with the exact (unregularised) KKT residual. */
typedef struct { REAL Q[9], N[9], G[9]; int r, nn, col0; } FootBasis;

static void mul_gamma(const FootBasis *fb, int nf, const REAL *v, REAL *out) {
  for (int f = 0; f < nf; ++f)
    for (int a = 0; a < 3; ++a)
      out[3 * f + a] = fb[f].G[a * 3] * v[3 * f] + fb[f].G[a * 3 + 1] * v[3 * f + 1] + fb[f].G[a * 3 + 2] * v[3 * f + 2];
}
/* out(n) = A_act^T y (active rows only) */
static void mul_At_act(const Port *s, const int *act, const REAL *y, REAL *out) {
  for (int f = 0; f < s->nf; ++f)
    for (int c = 0; c < 3; ++c) {
      REAL t = 0;
      for (int r = 0; r < 5; ++r) if (act[5 * f + r]) t += s->As[15 * f + 3 * r + c] * y[5 * f + r];
      out[3 * f + c] = t;
    }
}
static void kreg_apply(Port *s, const FootBasis *fb, const int *act, const REAL *Hinv_chol, int nw, const REAL *r1,
                       const REAL *r2, REAL *x, REAL *y, REAL *wk) {
  const int n = s->n, m = s->m, nf = s->nf;
  REAL *u = wk, *Pu = u + n, *tf = Pu + n, *v = tf + n, *Pv = v + n, *xN = Pv + n, *rw = xN + n, *Au = rw + n;
  const REAL dl = (REAL)Q_DELTA;
  mul_At_act(s, act, r2, v);
  mul_gamma(fb, nf, v, u);                       /* u = Gamma A^T r2 */
  mul_P(s, u, Pu);
  for (int i = 0; i < n; ++i) v[i] = r1[i] - Pu[i] - dl * u[i];
  mul_gamma(fb, nf, v, tf);                      /* tf */
  mul_P(s, tf, Pv);
  for (int i = 0; i < n; ++i) v[i] = r1[i] - Pu[i] - dl * Pv[i];
  for (int f = 0; f < nf; ++f)
    for (int k = 0; k < fb[f].nn; ++k)
      rw[fb[f].col0 + k] = fb[f].N[3 * k] * v[3 * f] + fb[f].N[3 * k + 1] * v[3 * f + 1] + fb[f].N[3 * k + 2] * v[3 * f + 2];
  chol_solve(Hinv_chol, nw, nw, rw);
  for (int f = 0; f < nf; ++f)
    for (int c = 0; c < 3; ++c) {
      REAL t = 0;
      for (int k = 0; k < fb[f].nn; ++k) t += fb[f].N[3 * k + c] * rw[fb[f].col0 + k];
      xN[3 * f + c] = t;
    }
  mul_P(s, xN, Pv);                               /* P xN */
  mul_gamma(fb, nf, Pv, v);                       /* Gamma P xN */
  for (int i = 0; i < n; ++i) x[i] = u[i] + dl * (tf[i] - v[i]) + xN[i];
  for (int i = 0; i < n; ++i) v[i] = r1[i] - Pu[i] - dl * u[i] - Pv[i];
  mul_gamma(fb, nf, v, tf);                       /* reuse tf = Gamma(...) */
  mul_A(s, tf, y);                                /* y = A Gamma(...) on all rows; mask below */
  mul_A(s, u, Au);
  for (int i = 0; i < m; ++i) y[i] = act[i] ? y[i] - (r2[i] - Au[i]) / dl : 0;
}

static int polish_reduced(Port *s) {
  const int n = s->n, m = s->m, nf = s->nf;
  const REAL dl = (REAL)Q_DELTA;
  int *act = (int *)calloc(m, sizeof(int));
  FootBasis *fb = (FootBasis *)calloc(nf, sizeof(FootBasis));
  for (int i = 0; i < m; ++i) act[i] = (s->z[i] - s->ls[i] < -s->y[i]) ? -1 : ((s->us[i] - s->z[i] < s->y[i]) ? 1 : 0);
  int nw = 0;
  for (int f = 0; f < nf; ++f) {
    const REAL *a = s->As + 15 * f;
    FootBasis *b = &fb[f];
    int r = 0;
    for (int row = 0; row < 5 && r < 3; ++row) {
      if (!act[5 * f + row]) continue;
      REAL v[3] = {a[row * 3], a[row * 3 + 1], a[row * 3 + 2]};
      REAL nrm0 = sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
      for (int pass = 0; pass < 2; ++pass)
        for (int k = 0; k < r; ++k) {
          REAL d = v[0] * b->Q[3 * k] + v[1] * b->Q[3 * k + 1] + v[2] * b->Q[3 * k + 2];
          v[0] -= d * b->Q[3 * k]; v[1] -= d * b->Q[3 * k + 1]; v[2] -= d * b->Q[3 * k + 2];
        }
      REAL nrm = sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
      if (!(nrm > (REAL)1e-6 * nrm0)) continue;
      b->Q[3 * r] = v[0] / nrm; b->Q[3 * r + 1] = v[1] / nrm; b->Q[3 * r + 2] = v[2] / nrm;
      ++r;
    }
    b->r = r;
    /* complement */
    REAL *N = b->N, *Q = b->Q;
    if (r == 0) { REAL I3[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}; memcpy(N, I3, sizeof I3); }
    else if (r == 1) {
      int imin = fabs(Q[0]) <= fabs(Q[1]) ? (fabs(Q[0]) <= fabs(Q[2]) ? 0 : 2) : (fabs(Q[1]) <= fabs(Q[2]) ? 1 : 2);
      REAL e[3] = {0, 0, 0}; e[imin] = 1;
      REAL d = Q[imin], v[3] = {e[0] - d * Q[0], e[1] - d * Q[1], e[2] - d * Q[2]};
      REAL nr = sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
      N[0] = v[0] / nr; N[1] = v[1] / nr; N[2] = v[2] / nr;
      N[3] = Q[1] * N[2] - Q[2] * N[1]; N[4] = Q[2] * N[0] - Q[0] * N[2]; N[5] = Q[0] * N[1] - Q[1] * N[0];
    } else if (r == 2) {
      N[0] = Q[1] * Q[5] - Q[2] * Q[4]; N[1] = Q[2] * Q[3] - Q[0] * Q[5]; N[2] = Q[0] * Q[4] - Q[1] * Q[3];
      REAL nr = sqrt(N[0] * N[0] + N[1] * N[1] + N[2] * N[2]);
      N[0] /= nr; N[1] /= nr; N[2] /= nr;
    }
    b->nn = 3 - r; b->col0 = nw; nw += b->nn;
    /* Gamma = Q (Q^T B Q)^{-1} Q^T, B = A_act^T A_act */
    REAL B[9] = {0}, G[9] = {0}, Gi[9] = {0};
    for (int row = 0; row < 5; ++row) if (act[5 * f + row])
      for (int c1 = 0; c1 < 3; ++c1) for (int c2 = 0; c2 < 3; ++c2) B[c1 * 3 + c2] += a[row * 3 + c1] * a[row * 3 + c2];
    for (int k1 = 0; k1 < r; ++k1) for (int k2 = 0; k2 < r; ++k2) {
      REAL t = 0;
      for (int c1 = 0; c1 < 3; ++c1) for (int c2 = 0; c2 < 3; ++c2) t += Q[3 * k1 + c1] * B[c1 * 3 + c2] * Q[3 * k2 + c2];
      G[k1 * 3 + k2] = t;
    }
    /* invert r x r SPD G by Gauss-Jordan */
    REAL Mx[18];
    for (int i = 0; i < r; ++i) for (int j = 0; j < r; ++j) { Mx[i * 6 + j] = G[i * 3 + j]; Mx[i * 6 + 3 + j] = (i == j); }
    for (int p = 0; p < r; ++p) {
      REAL d = Mx[p * 6 + p];
      for (int j = 0; j < 6; ++j) Mx[p * 6 + j] /= d;
      for (int i = 0; i < r; ++i) if (i != p) { REAL fct = Mx[i * 6 + p]; for (int j = 0; j < 6; ++j) Mx[i * 6 + j] -= fct * Mx[p * 6 + j]; }
    }
    for (int i = 0; i < r; ++i) for (int j = 0; j < r; ++j) Gi[i * 3 + j] = Mx[i * 6 + 3 + j];
    for (int c1 = 0; c1 < 3; ++c1) for (int c2 = 0; c2 < 3; ++c2) {
      REAL t = 0;
      for (int k1 = 0; k1 < r; ++k1) for (int k2 = 0; k2 < r; ++k2) t += Q[3 * k1 + c1] * Gi[k1 * 3 + k2] * Q[3 * k2 + c2];
      b->G[c1 * 3 + c2] = t;
    }
  }
  /* W = P N (n x nw); H_d = N^T W + delta I - delta W^T Gamma W */
  REAL *W = ralloc((size_t)n * (nw + 1)), *GW = ralloc((size_t)n * (nw + 1)), *Hd = ralloc((size_t)(nw + 1) * (nw + 1));
  for (int i = 0; i < n; ++i)
    for (int f = 0; f < nf; ++f)
      for (int k = 0; k < fb[f].nn; ++k) {
        const REAL *row = s->Ps + (size_t)i * n + 3 * f;
        W[(size_t)i * nw + fb[f].col0 + k] = row[0] * fb[f].N[3 * k] + row[1] * fb[f].N[3 * k + 1] + row[2] * fb[f].N[3 * k + 2];
      }
  const int no_corr = getenv("PORT_POLISH_CORR") == 0;   /* the O(delta) Schur correction is not needed (tests) */
  for (int f = 0; f < nf; ++f)
    for (int a = 0; a < 3; ++a)
      for (int w2 = 0; w2 < nw; ++w2)
        GW[(size_t)(3 * f + a) * nw + w2] = fb[f].G[a * 3] * W[(size_t)(3 * f) * nw + w2] + fb[f].G[a * 3 + 1] * W[(size_t)(3 * f + 1) * nw + w2] +
                                            fb[f].G[a * 3 + 2] * W[(size_t)(3 * f + 2) * nw + w2];
  for (int f1 = 0; f1 < nf; ++f1)
    for (int k1 = 0; k1 < fb[f1].nn; ++k1) {
      const int w1 = fb[f1].col0 + k1;
      for (int w2 = 0; w2 < nw; ++w2) {
        REAL t = fb[f1].N[3 * k1] * W[(size_t)(3 * f1) * nw + w2] + fb[f1].N[3 * k1 + 1] * W[(size_t)(3 * f1 + 1) * nw + w2] +
                 fb[f1].N[3 * k1 + 2] * W[(size_t)(3 * f1 + 2) * nw + w2];
        if (w1 == w2) t += dl;
        if (!no_corr) { REAL c = 0; for (int i = 0; i < n; ++i) c += W[(size_t)i * nw + w1] * GW[(size_t)i * nw + w2]; t -= dl * c; }
        Hd[(size_t)w1 * nw + w2] = t;
      }
    }
  int bad = 0;
  for (int j = 0; j < nw && !bad; ++j) {
    REAL d = Hd[(size_t)j * nw + j];
    for (int k = 0; k < j; ++k) d -= Hd[(size_t)j * nw + k] * Hd[(size_t)j * nw + k];
    if (!(d > 0)) { bad = 1; break; }
    d = sqrt(d); Hd[(size_t)j * nw + j] = d;
    for (int i = j + 1; i < nw; ++i) {
      REAL t = Hd[(size_t)i * nw + j];
      for (int k = 0; k < j; ++k) t -= Hd[(size_t)i * nw + k] * Hd[(size_t)j * nw + k];
      Hd[(size_t)i * nw + j] = t / d;
    }
  }
  int ok = 0;
  if (!bad) {
    s->nfact++;
    REAL *r1 = ralloc(n), *r2 = ralloc(m), *x = ralloc(n), *y = ralloc(m), *dx = ralloc(n), *dy = ralloc(m), *e1 = ralloc(n), *e2 = ralloc(m), *wk = ralloc(8 * n + m);
    for (int i = 0; i < n; ++i) r1[i] = -s->qs[i];
    for (int i = 0; i < m; ++i) r2[i] = act[i] < 0 ? s->ls[i] : (act[i] > 0 ? s->us[i] : 0);
    if (!getenv("PORT_POLISH_FULL")) {
      /* projected form: range part exact (u), iterative refinement only in the null space */
      REAL *u = wk, *Pu = u + n, *g = Pu + n, *xN = g + n, *PxN = xN + n, *rw = PxN + n, *wv = rw + n, *v = wv + n;
      mul_At_act(s, act, r2, v); mul_gamma(fb, nf, v, u); mul_P(s, u, Pu);
      for (int i = 0; i < n; ++i) { g[i] = r1[i] - Pu[i]; xN[i] = 0; PxN[i] = 0; wv[i] = 0; }
      for (int it = 0; it <= Q_POLISH_REFINE; ++it) {
        for (int f = 0; f < nf; ++f) for (int k = 0; k < fb[f].nn; ++k) {
          REAL t = 0; for (int c = 0; c < 3; ++c) t += fb[f].N[3 * k + c] * (g[3 * f + c] - PxN[3 * f + c]);
          rw[fb[f].col0 + k] = t; }
        chol_solve(Hd, nw, nw, rw);
        for (int k = 0; k < nw; ++k) wv[k] += rw[k];
        for (int f = 0; f < nf; ++f) for (int c = 0; c < 3; ++c) { REAL t = 0; for (int k = 0; k < fb[f].nn; ++k) t += fb[f].N[3 * k + c] * wv[fb[f].col0 + k]; xN[3 * f + c] = t; }
        mul_P(s, xN, PxN);
      }
      for (int i = 0; i < n; ++i) { x[i] = u[i] + xN[i]; v[i] = g[i] - PxN[i]; }
      mul_gamma(fb, nf, v, rw); mul_A(s, rw, y);
      for (int i = 0; i < m; ++i) if (!act[i]) y[i] = 0;
    } else {
    kreg_apply(s, fb, act, Hd, nw, r1, r2, x, y, wk);
    for (int it = 0; it < Q_POLISH_REFINE; ++it) {
      mul_P(s, x, e1); mul_At_act(s, act, y, s->tn); mul_A(s, x, e2);
      for (int i = 0; i < n; ++i) e1[i] = r1[i] - e1[i] - s->tn[i];
      for (int i = 0; i < m; ++i) e2[i] = act[i] ? r2[i] - e2[i] : 0;
      kreg_apply(s, fb, act, Hd, nw, e1, e2, dx, dy, wk);
      for (int i = 0; i < n; ++i) x[i] += dx[i];
      for (int i = 0; i < m; ++i) y[i] += dy[i];
    }
    }
    REAL *zpol = e2;
    mul_A(s, x, zpol);
    for (int i = 0; i < m; ++i) { REAL t = zpol[i] + y[i]; REAL zc = t < s->ls[i] ? s->ls[i] : (t > s->us[i] ? s->us[i] : t); zpol[i] = zc; y[i] = t - zc; }
    REAL pri, dua;
    update_info(s, x, zpol, y, &pri, &dua);
    ok = (pri < s->pri_res && dua < s->dua_res) || (pri < s->pri_res && s->dua_res < (REAL)1e-10) || (dua < s->dua_res && s->pri_res < (REAL)1e-10);
    if (ok) { s->pri_res = pri; s->dua_res = dua; memcpy(s->x, x, sizeof(REAL) * n); memcpy(s->z, zpol, sizeof(REAL) * m); memcpy(s->y, y, sizeof(REAL) * m); }
    free(r1); free(r2); free(x); free(y); free(dx); free(dy); free(e1); free(e2); free(wk);
  }
  s->status_polish = ok ? 1 : -1;
  free(W); free(GW); free(Hd); free(act); free(fb);
  return ok;
}

/*
 * One compute_contact_forces call.  in[] = flat record as doubles.  forces_out[12h] = -D x when
 * status == SOLVED (mpc_osqp.cc:788-790); returns 1 on success, 0 otherwise (reference returns []).
 * info[8] = {iter, status, status_polish, rho_updates, nfact, 0, 0, first_run}; dinfo[8] = {pri, dua, rho, 0, c}
 */
int port_solve(void *hd, const double *in, double *forces_out, int64_t *info, double *dinfo) {
  Port *s = (Port *)hd;
  const int n = s->n, m = s->m;
  mpc_assemble(&s->mdl, in, &s->wk, s->P, s->q, s->cone, s->l, s->u);
  for (int i = 0; i < m; ++i) { /* mpc_osqp.cc:720-721 */
    if (s->l[i] < -(REAL)Q_INFTY) s->l[i] = -(REAL)Q_INFTY;
    if (s->u[i] > (REAL)Q_INFTY) s->u[i] = (REAL)Q_INFTY;
  }
  const int first = s->first_run;
  s->nfact = 0;
  /* data -> scaled copies.  On later calls the equilibration sees the PREVIOUS q (osqp_update_P_A
   * runs scale_data before osqp_update_lin_cost delivers the new q). */
  memcpy(s->Ps, s->P, sizeof(REAL) * n * n);
  for (int f = 0; f < s->nf; ++f) memcpy(s->As + 15 * f, s->cone, sizeof(REAL) * 15);
  memcpy(s->qs, first ? s->q : s->q_old, sizeof(REAL) * n);
  scale_data(s);
  if (!first)                                                                   /* osqp.c:765-770 */
    for (int i = 0; i < n; ++i) s->qs[i] = (s->D[i] * s->q[i]) * s->c;
  for (int i = 0; i < m; ++i) { s->ls[i] = s->E[i] * s->l[i]; s->us[i] = s->E[i] * s->u[i]; }
  memcpy(s->q_old, s->q, sizeof(REAL) * n);
  if (first) {
    classify_rows(s, 1);                                                        /* osqp.c:205 set_rho_vec */
    if (factor_K(s)) return 0;
    s->first_run = 0;
  } else {
    /* update_P_A refactors with the old rho_vec; update_bounds refactors again only when a row type
     * changed (auxil.c:134-138).  One factorisation of the final matrix is equivalent. */
    classify_rows(s, 0);
    if (factor_K(s)) return 0;
  }
  s->status = ST_UNSOLVED; s->rho_updates = 0; s->status_polish = 0;            /* reset_info */

  int iter, checked = 0;
  for (iter = 1; iter <= Q_MAX_ITER; ++iter) {
    REAL *t;
    t = s->x; s->x = s->xp; s->xp = t;                                          /* osqp.c:356-357 */
    t = s->z; s->z = s->zp; s->zp = t;
    /* auxil.c:164-190 rhs + KKT solve in reduced form */
    for (int i = 0; i < m; ++i) s->tm[i] = s->rho_vec[i] * s->zp[i] - s->y[i];  /* R (z - y/rho) */
    mul_At(s, s->tm, s->xt);
    for (int i = 0; i < n; ++i) s->xt[i] += (REAL)Q_SIGMA * s->xp[i] - s->qs[i];
    chol_solve(s->K, n, n, s->xt);
    mul_A(s, s->xt, s->zt);
    for (int i = 0; i < n; ++i) {                                               /* auxil.c:192-205 */
      s->x[i] = (REAL)Q_ALPHA * s->xt[i] + ((REAL)1.0 - (REAL)Q_ALPHA) * s->xp[i];
      s->dx[i] = s->x[i] - s->xp[i];
    }
    for (int i = 0; i < m; ++i) {                                               /* auxil.c:207-233 */
      REAL zr = (REAL)Q_ALPHA * s->zt[i] + ((REAL)1.0 - (REAL)Q_ALPHA) * s->zp[i];
      REAL zn = zr + s->rho_inv[i] * s->y[i];
      zn = zn < s->ls[i] ? s->ls[i] : (zn > s->us[i] ? s->us[i] : zn);
      s->z[i] = zn;
      s->dy[i] = s->rho_vec[i] * (zr - zn);
      s->y[i] += s->dy[i];
    }
    checked = (iter % Q_CHECK == 0);
    if (checked) {
      update_info(s, s->x, s->z, s->y, &s->pri_res, &s->dua_res);
      s->iter = iter;
      if (check_termination(s, 0)) break;
      REAL rn = rho_estimate(s);                                                /* auxil.c:57-77 adapt_rho */
      if (rn > s->rho * (REAL)Q_ADAPT_TOL || rn < s->rho / (REAL)Q_ADAPT_TOL) {
        if (update_rho(s, rn)) return 0;
        s->rho_updates++;
      }
    }
  }
  if (!checked) { update_info(s, s->x, s->z, s->y, &s->pri_res, &s->dua_res); s->iter = iter - 1; check_termination(s, 0); }
  if (s->status == ST_UNSOLVED && !check_termination(s, 1)) s->status = ST_MAX_ITER; /* osqp.c:564-568 */
  if (s->status == ST_SOLVED) {
    if (getenv("PORT_POLISH_DEBUG")) {
      REAL *sx = ralloc(n), *sz = ralloc(m), *sy = ralloc(m), *kx = ralloc(n); REAL p0 = s->pri_res, d0 = s->dua_res;
      memcpy(sx, s->x, sizeof(REAL) * n); memcpy(sz, s->z, sizeof(REAL) * m); memcpy(sy, s->y, sizeof(REAL) * m);
      int ok1 = polish_kkt(s); memcpy(kx, s->x, sizeof(REAL) * n); REAL p1 = s->pri_res, d1 = s->dua_res;
      memcpy(s->x, sx, sizeof(REAL) * n); memcpy(s->z, sz, sizeof(REAL) * m); memcpy(s->y, sy, sizeof(REAL) * m); s->pri_res = p0; s->dua_res = d0;
      int ok2 = polish(s);
      REAL mx = 0, nx = 0; for (int i = 0; i < n; ++i) { REAL d = fabs(kx[i] - s->x[i]); if (d > mx) mx = d; if (fabs(s->x[i]) > nx) nx = fabs(s->x[i]); }
      fprintf(stderr, "polish dbg: kkt ok=%d pri %.3e dua %.3e | ns ok=%d pri %.3e dua %.3e | admm pri %.3e dua %.3e | dx %.3e / %.3e\n", ok1, (double)p1, (double)d1, ok2, (double)s->pri_res, (double)s->dua_res, (double)p0, (double)d0, (double)mx, (double)nx);
      free(sx); free(sz); free(sy); free(kx);
    } else {
      const char *pm = getenv("PORT_POLISH");   /* default: the form the HIP kernel runs */
      if (pm && !strcmp(pm, "kkt")) polish_kkt(s);            /* literal polish.c (dense LU), checker of the reduced form */
      else if (pm && !strcmp(pm, "exact")) polish(s);          /* delta -> 0 limit, for comparison only */
      else polish_reduced(s);
    }
  }
  int has_sol = !(s->status == ST_PRIMAL_INF || s->status == ST_PRIMAL_INF_INACC || s->status == ST_DUAL_INF ||
                  s->status == ST_DUAL_INF_INACC || s->status == ST_NON_CVX);
  if (!has_sol) { /* auxil.c:558-560 cold start for the next run */
    memset(s->x, 0, sizeof(REAL) * n); memset(s->z, 0, sizeof(REAL) * m); memset(s->y, 0, sizeof(REAL) * m);
  }
  if (info) { info[0] = s->iter; info[1] = s->status; info[2] = s->status_polish; info[3] = s->rho_updates;
              info[4] = s->nfact; info[5] = info[6] = 0; info[7] = first; }
  if (dinfo) { dinfo[0] = s->pri_res; dinfo[1] = s->dua_res; dinfo[2] = s->rho; dinfo[3] = 0; dinfo[4] = s->c;
               dinfo[5] = dinfo[6] = dinfo[7] = 0; }
  if (s->status != ST_SOLVED) return 0;
  for (int i = 0; i < n; ++i) forces_out[i] = -(double)(s->D[i] * s->x[i]);
  return 1;
}

void port_get_qp(void *hd, double *P, double *q, double *l, double *u) {
  Port *s = (Port *)hd;
  for (size_t i = 0; i < (size_t)s->n * s->n; ++i) P[i] = s->P[i];
  for (int i = 0; i < s->n; ++i) q[i] = s->q[i];
  for (int i = 0; i < s->m; ++i) { l[i] = s->l[i]; u[i] = s->u[i]; }
}
void port_get_state(void *hd, double *x, double *z, double *y, double *D, double *E, double *rho_c) {
  Port *s = (Port *)hd;
  for (int i = 0; i < s->n; ++i) { x[i] = s->x[i]; D[i] = s->D[i]; }
  for (int i = 0; i < s->m; ++i) { z[i] = s->z[i]; y[i] = s->y[i]; E[i] = s->E[i]; }
  rho_c[0] = s->rho; rho_c[1] = s->c;
}
int port_sizeof_real(void) { return (int)sizeof(REAL); }

typedef struct { void **handles; const double *in; double *forces; int64_t *info; int inlen, nout, lo, hi; } PJob;
static void *pworker(void *arg) {
  PJob *j = (PJob *)arg;
  for (int r = j->lo; r < j->hi; ++r) {
    int64_t inf[8];
    double *f = j->forces + (size_t)r * j->nout;
    if (!port_solve(j->handles[r], j->in + (size_t)r * j->inlen, f, inf, 0))
      for (int k = 0; k < j->nout; ++k) f[k] = NAN;
    if (j->info) memcpy(j->info + (size_t)r * 8, inf, sizeof inf);
  }
  return 0;
}
void port_batch_solve(void **handles, int N, int h, const double *in, double *forces, int64_t *info, int nthreads) {
  if (nthreads < 1) nthreads = 1;
  if (nthreads > 256) nthreads = 256;
  pthread_t th[256];
  PJob jobs[256];
  int per = (N + nthreads - 1) / nthreads;
  for (int t = 0; t < nthreads; ++t) {
    jobs[t] = (PJob){handles, in, forces, info, mpc_in_len(h), 12 * h, t * per, (t + 1) * per < N ? (t + 1) * per : N};
    if (nthreads == 1) pworker(&jobs[t]); else pthread_create(&th[t], 0, pworker, &jobs[t]);
  }
  if (nthreads > 1) for (int t = 0; t < nthreads; ++t) pthread_join(th[t], 0);
}
