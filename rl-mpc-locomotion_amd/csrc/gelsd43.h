// gelsd43.h -- the 4 x 3 single-precision least squares of the ground-normal fit, in the reference's own arithmetic.
//
// The reference solves  foot_history[4 x 3] n = 1  with scipy.linalg.lstsq on float32 (MPC_Controller/common/StateEstimator.py:130),
// i.e. LAPACK's SGELSD as scipy 1.15 ships it (OpenBLAS 0.3.28: reference LAPACK 3.11 Fortran built without FMA contraction, BLAS
// level-1/2 calls served by OpenBLAS' SkylakeX kernels).  A mathematically equivalent solve in another arithmetic lands 1e-7 .. 3e-7
// away, and OSQP at eps 1e-3 turns that into different discrete decisions (polish accepted / rejected, one more 25-iteration block),
// so this file walks SGELSD's own path for this one shape -- every operation in the order and precision the library executes it:
//
//   SGELSD (m = 4 >= mnthr = 4)  ->  SGEQR2 (three Householder reflectors)  ->  SORM2R (Q^T b)  ->  SGEBD2 on the 3 x 3 R
//   (tauq(1) = tauq(3) = taup(2) = 0 by structure)  ->  SORM2R (Q_b^T b)  ->  SLALSD (n <= smlsiz: scale by the max-norm, SLASDQ =
//   SBDSQR with VT and C = b followed by a re-sort into increasing order, threshold rcond * sigma_max, VT^T (c / sigma))  ->  SORML2 (P b).
//
// BLAS kernel facts pinned by experiment against the library (tests/test_gelsd43.py re-checks the end result against scipy when it
// is importable):  SNRM2 accumulates in double and rounds once;  SGEMV^T sums  m = 4, n = 2: (p0 + p1) + (p2 + p3);  m = 4, n = 1:
// ((p0 + p1) + p2) + p3;  m = 3: fma(a2, x2, fma(a0, x0, a1 x1));  m = 2: fma(a0, x0, a1 x1);  SGER / SAXPY: a += (alpha y_j) x_i
// as one fma;  SROT and SGEMM: see rot() / the VT^T product below.
// Single precision throughout; compile WITHOUT floating-point contraction (the pragma below); fmaf only where the library fuses.
#pragma once

#include <math.h>

#include "mpc_core.h"

namespace mpc {
namespace gelsd43 {

#if defined(__clang__)
#define GELSD_NO_CONTRACT _Pragma("clang fp contract(off)")
#else
#define GELSD_NO_CONTRACT
#endif

MPC_HD float sgn(float a, float b) { return b >= 0.f ? fabsf(a) : -fabsf(a); }      // Fortran SIGN(a, b)  (b = -0.0 does not occur on this path)

// SNRM2 (OpenBLAS nrm2_sse: squares and sum in double, one rounding at the end)
MPC_HD float nrm2(int n, const float *x, int inc) {
  GELSD_NO_CONTRACT
  double s = 0.0;
  for (int i = 0; i < n; ++i) s += (double)x[i * inc] * (double)x[i * inc];
  return (float)sqrt(s);
}
// SLAPY2 (LAPACK 3.10+)
MPC_HD float lapy2(float x, float y) {
  GELSD_NO_CONTRACT
  const float xa = fabsf(x), ya = fabsf(y), w = fmaxf(xa, ya), z = fminf(xa, ya);
  if (z == 0.f) return w;
  const float q = z / w;
  return w * sqrtf(1.f + q * q);
}
// SLARFG: n = 1 + number of entries of x
MPC_HD void larfg(int n, float &alpha, float *x, int incx, float &tau) {
  GELSD_NO_CONTRACT
  tau = 0.f;
  if (n <= 1) return;
  const float xnorm = nrm2(n - 1, x, incx);
  if (xnorm == 0.f) return;
  const float beta = -sgn(lapy2(alpha, xnorm), alpha);
  tau = (beta - alpha) / beta;
  const float sc = 1.f / (alpha - beta);
  for (int i = 0; i < n - 1; ++i) x[i * incx] = x[i * incx] * sc;      // SSCAL
  alpha = beta;
}
// SGEMV('T'), beta = 0: w[j] = sum_i C(i, j) v[i], i < m (C column-major, leading dimension ldc; v with stride incv)
MPC_HD void gemv_t(int m, int n, const float *C, int ldc, const float *v, int incv, float *w) {
  GELSD_NO_CONTRACT
  for (int j = 0; j < n; ++j) {
    const float *c = C + j * ldc;
    if (m == 1) w[j] = c[0] * v[0];
    else if (m == 2) w[j] = fmaf(c[0], v[0], c[1] * v[incv]);
    else if (m == 3) w[j] = fmaf(c[2], v[2 * incv], fmaf(c[0], v[0], c[1] * v[incv]));
    else if (n == 1) w[j] = ((c[0] * v[0] + c[1] * v[incv]) + c[2] * v[2 * incv]) + c[3] * v[3 * incv];
    else w[j] = (c[0] * v[0] + c[1] * v[incv]) + (c[2] * v[2 * incv] + c[3] * v[3 * incv]);
  }
}
// SGEMV('N'), beta = 0: w[i] = sum_j C(i, j) v[j], i < m, j < n
MPC_HD void gemv_n(int m, int n, const float *C, int ldc, const float *v, int incv, float *w) {
  GELSD_NO_CONTRACT
  for (int i = 0; i < m; ++i) {
    float acc = 0.f;
    for (int j = 0; j < n; ++j) acc = fmaf(C[i + j * ldc], v[j * incv], acc);
    w[i] = acc;
  }
}
// SGER: C(i, j) += alpha x[i] y[j]
MPC_HD void ger(int m, int n, float alpha, const float *x, int incx, const float *y, int incy, float *C, int ldc) {
  GELSD_NO_CONTRACT
  for (int j = 0; j < n; ++j) {
    const float t = alpha * y[j * incy];
    for (int i = 0; i < m; ++i) C[i + j * ldc] = fmaf(t, x[i * incx], C[i + j * ldc]);
  }
}
// SLARF('Left'): C (m x n) <- (I - tau v v^T) C, with the library's scan for trailing zeros of v and zero columns of C
MPC_HD void larf_left(int m, int n, const float *v, int incv, float tau, float *C, int ldc) {
  if (tau == 0.f) return;
  int lastv = m;
  while (lastv > 0 && v[(lastv - 1) * incv] == 0.f) --lastv;
  if (lastv == 0) return;
  int lastc = n;                                                     // ILASLC
  if (!(n == 0 || C[(n - 1) * ldc] != 0.f || C[lastv - 1 + (n - 1) * ldc] != 0.f)) {
    for (lastc = n; lastc >= 1; --lastc) {
      bool nz = false;
      for (int i = 0; i < lastv; ++i) nz = nz || C[i + (lastc - 1) * ldc] != 0.f;
      if (nz) break;
    }
  }
  if (lastc == 0) return;
  float w[3];
  gemv_t(lastv, lastc, C, ldc, v, incv, w);
  ger(lastv, lastc, -tau, v, incv, w, 1, C, ldc);
}
// SLARF('Right'): C (m x n) <- C (I - tau v v^T)
MPC_HD void larf_right(int m, int n, const float *v, int incv, float tau, float *C, int ldc) {
  if (tau == 0.f) return;
  int lastv = n;
  while (lastv > 0 && v[(lastv - 1) * incv] == 0.f) --lastv;
  if (lastv == 0) return;
  int lastc = m;                                                     // ILASLR
  if (!(m == 0 || C[m - 1] != 0.f || C[m - 1 + (lastv - 1) * ldc] != 0.f)) {
    lastc = 0;
    for (int j = 0; j < lastv; ++j) {
      int i = m;
      while (i >= 1 && C[(i > 1 ? i : 1) - 1 + j * ldc] == 0.f) --i;
      lastc = lastc > i ? lastc : i;
    }
  }
  if (lastc == 0) return;
  float w[3];
  gemv_n(lastc, lastv, C, ldc, v, incv, w);
  ger(lastc, lastv, -tau, w, 1, v, incv, C, ldc);
}
// SLARTG (LAPACK 3.10+, la_xisnan-free branch; the scaled branch serves |f| or |g| outside (rtmin, rtmax))
MPC_HD void lartg(float f, float g, float &c, float &s, float &r) {
  GELSD_NO_CONTRACT
  const float safmin = 1.17549435e-38f, safmax = 8.50705917e+37f;      // 2^-126, 2^126
  const float rtmin = 1.08420217e-19f;                                    // sqrt(safmin) = 2^-63
  const float rtmax = 6.52190267e+18f;                                    // sqrt(safmax / 2)
  const float f1 = fabsf(f), g1 = fabsf(g);
  if (g == 0.f) { c = 1.f; s = 0.f; r = f; }
  else if (f == 0.f) { c = 0.f; s = sgn(1.f, g); r = g1; }
  else if (f1 > rtmin && f1 < rtmax && g1 > rtmin && g1 < rtmax) {
    const float d = sqrtf(f * f + g * g);
    c = f1 / d;
    r = sgn(d, f);
    s = g / r;
  } else {
    const float u = fminf(safmax, fmaxf(safmin, fmaxf(f1, g1)));
    const float fs = f / u, gs = g / u;
    const float d = sqrtf(fs * fs + gs * gs);
    c = fabsf(fs) / d;
    r = sgn(d, f);
    s = gs / r;
    r = r * u;
  }
}
// SLAS2: singular values of [[f, g], [0, h]]
MPC_HD void las2(float f, float g, float h, float &ssmin, float &ssmax) {
  GELSD_NO_CONTRACT
  const float fa = fabsf(f), ga = fabsf(g), ha = fabsf(h), fhmn = fminf(fa, ha), fhmx = fmaxf(fa, ha);
  if (fhmn == 0.f) {
    ssmin = 0.f;
    if (fhmx == 0.f) ssmax = ga;
    else { const float mx = fmaxf(fhmx, ga), mn = fminf(fhmx, ga), q = mn / mx; ssmax = mx * sqrtf(1.f + q * q); }
  } else if (ga < fhmx) {
    const float as = 1.f + fhmn / fhmx, at = (fhmx - fhmn) / fhmx, q = ga / fhmx, au = q * q;
    const float c = 2.f / (sqrtf(as * as + au) + sqrtf(at * at + au));
    ssmin = fhmn * c;
    ssmax = fhmx / c;
  } else {
    const float au = fhmx / ga;
    if (au == 0.f) { ssmin = (fhmn * fhmx) / ga; ssmax = ga; }
    else {
      const float as = 1.f + fhmn / fhmx, at = (fhmx - fhmn) / fhmx, p1 = as * au, p2 = at * au;
      const float c = 1.f / (sqrtf(1.f + p1 * p1) + sqrtf(1.f + p2 * p2));
      ssmin = (fhmn * c) * au;
      ssmin = ssmin + ssmin;
      ssmax = ga / (c + c);
    }
  }
}
// SLASV2: SVD of [[f, g], [0, h]]
MPC_HD void lasv2(float f, float g, float h, float &ssmin, float &ssmax, float &snr, float &csr, float &snl, float &csl) {
  GELSD_NO_CONTRACT
  const float eps = 5.96046448e-08f;      // SLAMCH('EPS') = 2^-24
  float ft = f, fa = fabsf(ft), ht = h, ha = fabsf(h);
  int pmax = 1;
  const bool swap = ha > fa;
  if (swap) { pmax = 3; float t = ft; ft = ht; ht = t; t = fa; fa = ha; ha = t; }
  const float gt = g, ga = fabsf(gt);
  float clt, crt, slt, srt;
  if (ga == 0.f) { ssmin = ha; ssmax = fa; clt = 1.f; crt = 1.f; slt = 0.f; srt = 0.f; }
  else {
    bool gasmal = true;
    if (ga > fa) {
      pmax = 2;
      if (fa / ga < eps) {
        gasmal = false;
        ssmax = ga;
        if (ha > 1.f) ssmin = fa / (ga / ha); else ssmin = (fa / ga) * ha;
        clt = 1.f; slt = ht / gt; srt = 1.f; crt = ft / gt;
      }
    }
    if (gasmal) {
      const float d = fa - ha;
      float l = (d == fa) ? 1.f : d / fa;
      const float m = gt / ft;
      float t = 2.f - l;
      const float mm = m * m, tt = t * t;
      const float s = sqrtf(tt + mm);
      const float r = (l == 0.f) ? fabsf(m) : sqrtf(l * l + mm);
      const float a = 0.5f * (s + r);
      ssmin = ha / a;
      ssmax = fa * a;
      if (mm == 0.f) {
        if (l == 0.f) t = sgn(2.f, ft) * sgn(1.f, gt);
        else t = gt / sgn(d, ft) + m / t;
      } else {
        t = (m / (s + t) + m / (r + l)) * (1.f + a);
      }
      l = sqrtf(t * t + 4.f);
      crt = 2.f / l;
      srt = t / l;
      clt = (crt + srt * m) / a;
      slt = (ht / ft) * srt / a;
    }
  }
  if (swap) { csl = srt; snl = crt; csr = slt; snr = clt; } else { csl = clt; snl = slt; csr = crt; snr = srt; }
  float tsign;
  if (pmax == 1) tsign = sgn(1.f, csr) * sgn(1.f, csl) * sgn(1.f, f);
  else if (pmax == 2) tsign = sgn(1.f, snr) * sgn(1.f, csl) * sgn(1.f, g);
  else tsign = sgn(1.f, snr) * sgn(1.f, snl) * sgn(1.f, h);
  ssmax = sgn(ssmax, tsign);
  ssmin = sgn(ssmin, tsign * sgn(1.f, f) * sgn(1.f, h));
}
// SROT on n entries: x <- c x + s y, y <- c y - s x
MPC_HD void rot(int n, float *x, int incx, float *y, int incy, float c, float s) {
  GELSD_NO_CONTRACT
  for (int i = 0; i < n; ++i) {
    const float xv = x[i * incx], yv = y[i * incy];
    x[i * incx] = fmaf(c, xv, s * yv);
    y[i * incy] = fmaf(c, yv, -(s * xv));
  }
}
// SLASR('L', 'V', dir): plane rotations (cs[j], sn[j]) between rows j and j + 1 of A (rows x ncol, leading dimension lda), forward or backward
MPC_HD void lasr_lv(bool forward, int rows, int ncol, const float *cs, const float *sn, float *A, int lda) {
  GELSD_NO_CONTRACT
  for (int jj = 0; jj < rows - 1; ++jj) {
    const int j = forward ? jj : rows - 2 - jj;
    const float ct = cs[j], st = sn[j];
    if (ct != 1.f || st != 0.f)
      for (int i = 0; i < ncol; ++i) {
        const float temp = A[j + 1 + i * lda];
        A[j + 1 + i * lda] = ct * temp - st * A[j + i * lda];
        A[j + i * lda] = st * temp + ct * A[j + i * lda];
      }
  }
}
// SBDSQR('U', 3, ncvt = 3, nru = 0, ncc = 1): singular values of the upper bidiagonal (d, e), VT <- rotations applied to the rows of
// the 3 x 3 identity, c <- left rotations applied to c.  Returns 0 (LAPACK info; > 0: no convergence within 6 n^2 sweeps).
MPC_HD int bdsqr3(float *d, float *e, float *vt /* 3 x 3 column-major */, float *c) {
  GELSD_NO_CONTRACT
  const int n = 3, maxitr = 6;
  const float eps = 5.96046448e-08f, unfl = 1.17549435e-38f;
  const float tolmul = 10.f;                                               // max(10, min(100, eps^(-1/8))) with eps^(-1/8) = 2^3
  const float tol = tolmul * eps;
  float smax = 0.f;
  for (int i = 0; i < n; ++i) smax = fmaxf(smax, fabsf(d[i]));
  for (int i = 0; i < n - 1; ++i) smax = fmaxf(smax, fabsf(e[i]));
  float smin = 0.f;
  float sminoa = fabsf(d[0]);
  if (sminoa != 0.f) {
    float mu = sminoa;
    for (int i = 1; i < n; ++i) {
      mu = fabsf(d[i]) * (mu / (mu + fabsf(e[i - 1])));
      sminoa = fminf(sminoa, mu);
      if (sminoa == 0.f) break;
    }
  }
  sminoa = sminoa / sqrtf((float)n);
  const float thresh = fmaxf(tol * sminoa, (float)maxitr * ((float)n * ((float)n * unfl)));
  const int maxitdivn = maxitr * n;
  int iterdivn = 0, iter = -1, oldll = -1, oldm = -1, m = n, idir = 0;       // m, ll: 1-based as in the Fortran
  float work[4 * 2];                                                           // cs / sn / oldcs / oldsn of a sweep (n - 1 = 2 each)
  float *w1 = work, *w2 = work + 2, *w3 = work + 4, *w4 = work + 6;
#define D(i) d[(i) - 1]
#define E(i) e[(i) - 1]
  for (;;) {
    if (m <= 1) break;
    if (iter >= n) { iter -= n; ++iterdivn; if (iterdivn >= maxitdivn) return 1; }
    smax = fabsf(D(m));
    int ll = 0;
    bool split = false;
    for (int lll = 1; lll <= m - 1; ++lll) {
      ll = m - lll;
      const float abss = fabsf(D(ll)), abse = fabsf(E(ll));
      if (abse <= thresh) { split = true; break; }
      smax = fmaxf(smax, fmaxf(abss, abse));
    }
    if (split) {
      E(ll) = 0.f;
      if (ll == m - 1) { m = m - 1; continue; }
    } else ll = 0;
    ll = ll + 1;
    if (ll == m - 1) {        // 2 x 2 block
      float sigmn, sigmx, sinr, cosr, sinl, cosl;
      lasv2(D(m - 1), E(m - 1), D(m), sigmn, sigmx, sinr, cosr, sinl, cosl);
      D(m - 1) = sigmx; E(m - 1) = 0.f; D(m) = sigmn;
      rot(3, vt + (m - 2), 3, vt + (m - 1), 3, cosr, sinr);
      rot(1, c + (m - 2), 1, c + (m - 1), 1, cosl, sinl);
      m = m - 2;
      continue;
    }
    if (ll > oldm || m < oldll) idir = (fabsf(D(ll)) >= fabsf(D(m))) ? 1 : 2;
    bool again = false;
    if (idir == 1) {
      if (fabsf(E(m - 1)) <= fabsf(tol) * fabsf(D(m))) { E(m - 1) = 0.f; continue; }
      float mu = fabsf(D(ll));
      smin = mu;
      for (int lll = ll; lll <= m - 1; ++lll) {
        if (fabsf(E(lll)) <= tol * mu) { E(lll) = 0.f; again = true; break; }
        mu = fabsf(D(lll + 1)) * (mu / (mu + fabsf(E(lll))));
        smin = fminf(smin, mu);
      }
    } else {
      if (fabsf(E(ll)) <= fabsf(tol) * fabsf(D(ll))) { E(ll) = 0.f; continue; }
      float mu = fabsf(D(m));
      smin = mu;
      for (int lll = m - 1; lll >= ll; --lll) {
        if (fabsf(E(lll)) <= tol * mu) { E(lll) = 0.f; again = true; break; }
        mu = fabsf(D(lll)) * (mu / (mu + fabsf(E(lll))));
        smin = fminf(smin, mu);
      }
    }
    if (again) continue;
    oldll = ll; oldm = m;
    float shift, r;
    if ((float)n * tol * (smin / smax) <= fmaxf(eps, 0.01f * tol)) shift = 0.f;
    else {
      float sll;
      if (idir == 1) { sll = fabsf(D(ll)); las2(D(m - 1), E(m - 1), D(m), shift, r); }
      else { sll = fabsf(D(m)); las2(D(ll), E(ll), D(ll + 1), shift, r); }
      if (sll > 0.f) { const float q = shift / sll; if (q * q < eps) shift = 0.f; }
    }
    iter = iter + m - ll;
    const int rows = m - ll + 1;
    if (shift == 0.f) {
      if (idir == 1) {
        float cs = 1.f, oldcs = 1.f, sn = 0.f, oldsn = 0.f;
        for (int i = ll; i <= m - 1; ++i) {
          lartg(D(i) * cs, E(i), cs, sn, r);
          if (i > ll) E(i - 1) = oldsn * r;
          lartg(oldcs * r, D(i + 1) * sn, oldcs, oldsn, D(i));
          w1[i - ll] = cs; w2[i - ll] = sn; w3[i - ll] = oldcs; w4[i - ll] = oldsn;
        }
        const float h = D(m) * cs;
        D(m) = h * oldcs;
        E(m - 1) = h * oldsn;
        lasr_lv(true, rows, 3, w1, w2, vt + (ll - 1), 3);
        lasr_lv(true, rows, 1, w3, w4, c + (ll - 1), 3);
        if (fabsf(E(m - 1)) <= thresh) E(m - 1) = 0.f;
      } else {
        float cs = 1.f, oldcs = 1.f, sn = 0.f, oldsn = 0.f;
        for (int i = m; i >= ll + 1; --i) {
          lartg(D(i) * cs, E(i - 1), cs, sn, r);
          if (i < m) E(i) = oldsn * r;
          lartg(oldcs * r, D(i - 1) * sn, oldcs, oldsn, D(i));
          w1[i - ll - 1] = cs; w2[i - ll - 1] = -sn; w3[i - ll - 1] = oldcs; w4[i - ll - 1] = -oldsn;
        }
        const float h = D(ll) * cs;
        D(ll) = h * oldcs;
        E(ll) = h * oldsn;
        lasr_lv(false, rows, 3, w3, w4, vt + (ll - 1), 3);
        lasr_lv(false, rows, 1, w1, w2, c + (ll - 1), 3);
        if (fabsf(E(ll)) <= thresh) E(ll) = 0.f;
      }
    } else {
      if (idir == 1) {
        float f = (fabsf(D(ll)) - shift) * (sgn(1.f, D(ll)) + shift / D(ll));
        float g = E(ll);
        for (int i = ll; i <= m - 1; ++i) {
          float cosr, sinr, cosl, sinl;
          lartg(f, g, cosr, sinr, r);
          if (i > ll) E(i - 1) = r;
          f = cosr * D(i) + sinr * E(i);
          E(i) = cosr * E(i) - sinr * D(i);
          g = sinr * D(i + 1);
          D(i + 1) = cosr * D(i + 1);
          lartg(f, g, cosl, sinl, r);
          D(i) = r;
          f = cosl * E(i) + sinl * D(i + 1);
          D(i + 1) = cosl * D(i + 1) - sinl * E(i);
          if (i < m - 1) { g = sinl * E(i + 1); E(i + 1) = cosl * E(i + 1); }
          w1[i - ll] = cosr; w2[i - ll] = sinr; w3[i - ll] = cosl; w4[i - ll] = sinl;
        }
        E(m - 1) = f;
        lasr_lv(true, rows, 3, w1, w2, vt + (ll - 1), 3);
        lasr_lv(true, rows, 1, w3, w4, c + (ll - 1), 3);
        if (fabsf(E(m - 1)) <= thresh) E(m - 1) = 0.f;
      } else {
        float f = (fabsf(D(m)) - shift) * (sgn(1.f, D(m)) + shift / D(m));
        float g = E(m - 1);
        for (int i = m; i >= ll + 1; --i) {
          float cosr, sinr, cosl, sinl;
          lartg(f, g, cosr, sinr, r);
          if (i < m) E(i) = r;
          f = cosr * D(i) + sinr * E(i - 1);
          E(i - 1) = cosr * E(i - 1) - sinr * D(i);
          g = sinr * D(i - 1);
          D(i - 1) = cosr * D(i - 1);
          lartg(f, g, cosl, sinl, r);
          D(i) = r;
          f = cosl * E(i - 1) + sinl * D(i - 1);
          D(i - 1) = cosl * D(i - 1) - sinl * E(i - 1);
          if (i > ll + 1) { g = sinl * E(i - 2); E(i - 2) = cosl * E(i - 2); }
          w1[i - ll - 1] = cosr; w2[i - ll - 1] = -sinr; w3[i - ll - 1] = cosl; w4[i - ll - 1] = -sinl;
        }
        E(ll) = f;
        if (fabsf(E(ll)) <= thresh) E(ll) = 0.f;
        lasr_lv(false, rows, 3, w3, w4, vt + (ll - 1), 3);
        lasr_lv(false, rows, 1, w1, w2, c + (ll - 1), 3);
      }
    }
  }
  for (int i = 1; i <= n; ++i)
    if (D(i) < 0.f) {
      D(i) = -D(i);
      for (int k = 0; k < 3; ++k) vt[(i - 1) + 3 * k] = -1.f * vt[(i - 1) + 3 * k];
    }
  for (int i = 1; i <= n - 1; ++i) {      // decreasing order
    int isub = 1;
    float sm = D(1);
    for (int j = 2; j <= n + 1 - i; ++j)
      if (D(j) <= sm) { isub = j; sm = D(j); }
    if (isub != n + 1 - i) {
      D(isub) = D(n + 1 - i);
      D(n + 1 - i) = sm;
      for (int k = 0; k < 3; ++k) { const float t = vt[(isub - 1) + 3 * k]; vt[(isub - 1) + 3 * k] = vt[(n - i) + 3 * k]; vt[(n - i) + 3 * k] = t; }
      const float t = c[isub - 1]; c[isub - 1] = c[n - i]; c[n - i] = t;
    }
  }
#undef D
#undef E
  return 0;
}
// SLASCL('G') for values in the normal range: one rounded quotient, then one product per entry
MPC_HD float lascl_mul(float cfrom, float cto) { return cto / cfrom; }

// x (3) = argmin |A x - 1|, A = 4 x 3 row-major float32 (the foot-contact history).  Returns LAPACK's rank.
MPC_HD int solve_ones(const float *A_rowmajor, float *x, float *dbg = nullptr) {
  GELSD_NO_CONTRACT
  const int lda = 4;
  float a[12], b[4] = {1.f, 1.f, 1.f, 1.f}, tau[3];
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 3; ++j) a[i + lda * j] = A_rowmajor[3 * i + j];
  // SGEQR2 + SORM2R('L', 'T')
  for (int i = 0; i < 3; ++i) {
    larfg(4 - i, a[i + lda * i], a + (i + 1) + lda * i, 1, tau[i]);
    const float aii = a[i + lda * i];
    a[i + lda * i] = 1.f;
    if (i < 2) larf_left(4 - i, 2 - i, a + i + lda * i, 1, tau[i], a + i + lda * (i + 1), lda);
    a[i + lda * i] = aii;
  }
  for (int i = 0; i < 3; ++i) {
    const float aii = a[i + lda * i];
    a[i + lda * i] = 1.f;
    larf_left(4 - i, 1, a + i + lda * i, 1, tau[i], b + i, 4);
    a[i + lda * i] = aii;
  }
  if (dbg) { for (int k = 0; k < 12; ++k) dbg[k] = a[k]; for (int k = 0; k < 3; ++k) dbg[12 + k] = tau[k]; for (int k = 0; k < 4; ++k) dbg[15 + k] = b[k]; }
  a[1] = a[2] = a[2 + lda] = 0.f;                                  // SLASET below the diagonal of R
  // SGEBD2 on the 3 x 3 R (lda = 4)
  float d[3], e[2], tauq[3], taup[3];
  for (int i = 0; i < 3; ++i) {
    larfg(3 - i, a[i + lda * i], a + (i + 1 < 3 ? i + 1 : 2) + lda * i, 1, tauq[i]);
    d[i] = a[i + lda * i];
    a[i + lda * i] = 1.f;
    if (i < 2) larf_left(3 - i, 2 - i, a + i + lda * i, 1, tauq[i], a + i + lda * (i + 1), lda);
    a[i + lda * i] = d[i];
    if (i < 2) {
      larfg(2 - i, a[i + lda * (i + 1)], a + i + lda * (i + 2 < 3 ? i + 2 : 2), lda, taup[i]);
      e[i] = a[i + lda * (i + 1)];
      a[i + lda * (i + 1)] = 1.f;
      larf_right(2 - i, 2 - i, a + i + lda * (i + 1), lda, taup[i], a + (i + 1) + lda * (i + 1), lda);
      a[i + lda * (i + 1)] = e[i];
    } else taup[i] = 0.f;
  }
  // SORMBR('Q', 'L', 'T') = SORM2R with the tauq reflectors
  for (int i = 0; i < 3; ++i) {
    const float aii = a[i + lda * i];
    a[i + lda * i] = 1.f;
    larf_left(3 - i, 1, a + i + lda * i, 1, tauq[i], b + i, 4);
    a[i + lda * i] = aii;
  }
  if (dbg) { for (int k = 0; k < 12; ++k) dbg[20 + k] = a[k]; for (int k = 0; k < 3; ++k) { dbg[32 + k] = d[k]; dbg[37 + k] = tauq[k]; dbg[40 + k] = taup[k]; dbg[43 + k] = b[k]; } dbg[35] = e[0]; dbg[36] = e[1]; }
  // SLALSD('U', n = 3 <= smlsiz)
  const float rcnd = 1.1920929e-07f;                               // scipy passes cond = finfo(float32).eps
  float orgnrm = 0.f;
  for (int i = 0; i < 3; ++i) orgnrm = fmaxf(orgnrm, fabsf(d[i]));
  for (int i = 0; i < 2; ++i) orgnrm = fmaxf(orgnrm, fabsf(e[i]));
  int rank = 0;
  if (orgnrm == 0.f) { x[0] = x[1] = x[2] = 0.f; return 0; }
  {
    const float mul = lascl_mul(orgnrm, 1.f);
    if (mul != 1.f) { for (int i = 0; i < 3; ++i) d[i] = d[i] * mul; for (int i = 0; i < 2; ++i) e[i] = e[i] * mul; }
  }
  float vt[9] = {1.f, 0.f, 0.f, 0.f, 1.f, 0.f, 0.f, 0.f, 1.f};
  if (bdsqr3(d, e, vt, b) != 0) { x[0] = x[1] = x[2] = 0.f; return -1; }
  for (int i = 0; i < 3; ++i) {                                    // SLASDQ re-sorts what SBDSQR left in decreasing order into INCREASING order
    int isub = i;
    float sm = d[i];
    for (int j = i + 1; j < 3; ++j)
      if (d[j] < sm) { isub = j; sm = d[j]; }
    if (isub != i) {
      d[isub] = d[i]; d[i] = sm;
      for (int k = 0; k < 3; ++k) { const float t = vt[isub + 3 * k]; vt[isub + 3 * k] = vt[i + 3 * k]; vt[i + 3 * k] = t; }
      const float t = b[isub]; b[isub] = b[i]; b[i] = t;
    }
  }
  float dmax = 0.f;
  for (int i = 0; i < 3; ++i) dmax = fmaxf(dmax, fabsf(d[i]));
  const float tol = rcnd * dmax;
  for (int i = 0; i < 3; ++i) {
    if (d[i] <= tol) b[i] = 0.f;
    else { const float mul = lascl_mul(d[i], 1.f); if (mul != 1.f) b[i] = b[i] * mul; ++rank; }
  }
  float y[3];
  for (int i = 0; i < 3; ++i) y[i] = fmaf(vt[2 + 3 * i], b[2], fmaf(vt[1 + 3 * i], b[1], vt[0 + 3 * i] * b[0]));      // SGEMM('T', 'N', 3, 1, 3)
  {
    const float mul = lascl_mul(orgnrm, 1.f);
    if (mul != 1.f) for (int i = 0; i < 3; ++i) y[i] = y[i] * mul;
  }
  for (int i = 0; i < 3; ++i) b[i] = y[i];
  if (dbg) for (int k = 0; k < 3; ++k) dbg[46 + k] = b[k];
  // SORMBR('P', 'L', 'N') = SORML2('L', 'T', 2, 1, 2) on b(2:3): reflectors taup(2) (= 0), then taup(1)
  {
    const float aii = a[lda * 1];
    a[lda * 1] = 1.f;
    larf_left(2, 1, a + lda * 1, lda, taup[0], b + 1, 4);
    a[lda * 1] = aii;
  }
  x[0] = b[0]; x[1] = b[1]; x[2] = b[2];
  return rank;
}

}  // namespace gelsd43
}  // namespace mpc
