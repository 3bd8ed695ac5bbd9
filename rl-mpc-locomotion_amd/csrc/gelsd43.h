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
// BLAS kernel facts pinned by experiment against the library (tools/gelsd43/pin.py, output: profiles/r05_gelsd43_pinning.txt; tests/test_gelsd43.py re-checks the end result against scipy when it
// is importable):  SNRM2 accumulates in double and rounds once;  SGEMV^T sums  m = 4, n = 2: (p0 + p1) + (p2 + p3);  m = 4, n = 1:
// ((p0 + p1) + p2) + p3;  m = 3: fma(a2, x2, fma(a0, x0, a1 x1));  m = 2: fma(a0, x0, a1 x1);  SGER / SAXPY: a += (alpha y_j) x_i
// as one fma;  SROT and SGEMM: see rot() / the VT^T product below.
// Single precision throughout; compile WITHOUT floating-point contraction -- the pragma in every function AND -ffp-contract=off for the translation unit
// (csrc/Makefile: the pragma alone lets an fmaf whose multiplier constant-folds to 1 fuse with a neighbour on the device); fmaf only where the library fuses.
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
template <int N>
MPC_HD float nrm2(const float *x, int inc) {
  GELSD_NO_CONTRACT
  double s = 0.0;
#pragma unroll
  for (int i = 0; i < N; ++i) s += (double)x[i * inc] * (double)x[i * inc];
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
// SLARFG on a vector of N entries: alpha and the N - 1 entries of x
template <int N>
MPC_HD void larfg(float &alpha, float *x, int incx, float &tau) {
  GELSD_NO_CONTRACT
  tau = 0.f;
  if (N <= 1) return;
  const float xnorm = nrm2<(N > 1 ? N - 1 : 1)>(x, incx);
  if (xnorm == 0.f) return;
  const float beta = -sgn(lapy2(alpha, xnorm), alpha);
  tau = (beta - alpha) / beta;
  const float sc = 1.f / (alpha - beta);
#pragma unroll
  for (int i = 0; i < N - 1; ++i) x[i * incx] = x[i * incx] * sc;      // SSCAL
  alpha = beta;
}
// Everything below is written with compile-time loop bounds and constant array indices (template maxima, run-time extents as predicates):
// after inlining, the 4 x 3 matrix, the right-hand side and the 3 x 3 VT live in registers on the device -- no scratch memory.

// one column of SGEMV('T'), beta = 0: sum_{i < m} c[i] v[i * incv] in the order OpenBLAS' kernel uses for that m (header comment)
template <int M>
MPC_HD float gemv_t_col(int m, bool one_column, const float *c, const float *v, int incv) {
  GELSD_NO_CONTRACT
  if (m == 1) return c[0] * v[0];
  if (M >= 2 && m == 2) return fmaf(c[0], v[0], c[M >= 2 ? 1 : 0] * v[(M >= 2 ? 1 : 0) * incv]);
  if (M >= 3 && m == 3) return fmaf(c[M >= 3 ? 2 : 0], v[(M >= 3 ? 2 : 0) * incv], fmaf(c[0], v[0], c[M >= 2 ? 1 : 0] * v[(M >= 2 ? 1 : 0) * incv]));
  if (M >= 4) {
    const float p0 = c[0] * v[0], p1 = c[M >= 2 ? 1 : 0] * v[(M >= 2 ? 1 : 0) * incv], p2 = c[M >= 3 ? 2 : 0] * v[(M >= 3 ? 2 : 0) * incv],
                p3 = c[M >= 4 ? 3 : 0] * v[(M >= 4 ? 3 : 0) * incv];
    return one_column ? ((p0 + p1) + p2) + p3 : (p0 + p1) + (p2 + p3);
  }
  return 0.f;
}
// SLARF('Left'): C (m x n, m <= M, n <= N) <- (I - tau v v^T) C, with the library's scan for trailing zeros of v and zero columns of C;
// SGEMV('T') then SGER (a += (alpha y_j) x_i as one fma)
template <int M, int N>
MPC_HD void larf_left(const float *v, int incv, float tau, float *C, int ldc) {
  GELSD_NO_CONTRACT
  if (tau == 0.f) return;
  int lastv = M;
#pragma unroll
  for (int i = M - 1; i >= 0; --i)
    if (lastv == i + 1 && v[i * incv] == 0.f) lastv = i;
  if (lastv == 0) return;
  // ILASLC(lastv, N, C): the last column with a non-zero among its first lastv rows
  int lastc = 0;
#pragma unroll
  for (int j = 0; j < N; ++j) {
    bool nz = false;
#pragma unroll
    for (int i = 0; i < M; ++i) nz = nz || (i < lastv && C[i + j * ldc] != 0.f);
    if (nz) lastc = j + 1;
  }
  if (lastc == 0) return;
  float w[N];
#pragma unroll
  for (int j = 0; j < N; ++j) w[j] = j < lastc ? gemv_t_col<M>(lastv, lastc == 1, C + j * ldc, v, incv) : 0.f;
#pragma unroll
  for (int j = 0; j < N; ++j) {
    const float t = -tau * w[j];
#pragma unroll
    for (int i = 0; i < M; ++i)
      if (j < lastc && i < lastv) C[i + j * ldc] = fmaf(t, v[i * incv], C[i + j * ldc]);
  }
}
// SLARF('Right'): C (m x n) <- C (I - tau v v^T); SGEMV('N') (a sequential fma chain per row) then SGER
template <int M, int N>
MPC_HD void larf_right(const float *v, int incv, float tau, float *C, int ldc) {
  GELSD_NO_CONTRACT
  if (tau == 0.f) return;
  int lastv = N;
#pragma unroll
  for (int j = N - 1; j >= 0; --j)
    if (lastv == j + 1 && v[j * incv] == 0.f) lastv = j;
  if (lastv == 0) return;
  // ILASLR(M, lastv, C): the last row with a non-zero among its first lastv columns
  int lastc = 0;
#pragma unroll
  for (int i = 0; i < M; ++i) {
    bool nz = false;
#pragma unroll
    for (int j = 0; j < N; ++j) nz = nz || (j < lastv && C[i + j * ldc] != 0.f);
    if (nz) lastc = i + 1;
  }
  if (lastc == 0) return;
  float w[M];
#pragma unroll
  for (int i = 0; i < M; ++i) {
    float acc = 0.f;
#pragma unroll
    for (int j = 0; j < N; ++j)
      if (j < lastv) acc = fmaf(C[i + j * ldc], v[j * incv], acc);
    w[i] = acc;
  }
#pragma unroll
  for (int j = 0; j < N; ++j) {
    const float t = -tau * v[j * incv];
#pragma unroll
    for (int i = 0; i < M; ++i)
      if (j < lastv && i < lastc) C[i + j * ldc] = fmaf(t, w[i], C[i + j * ldc]);
  }
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
// SROT on the entries i < N of two strided vectors: x <- c x + s y, y <- c y - s x
template <int N>
MPC_HD void rot(float *x, int incx, float *y, int incy, float c, float s) {
  GELSD_NO_CONTRACT
#pragma unroll
  for (int i = 0; i < N; ++i) {
    const float xv = x[i * incx], yv = y[i * incy];
    x[i * incx] = fmaf(c, xv, s * yv);
    y[i * incy] = fmaf(c, yv, -(s * xv));
  }
}
// SLASR('L', 'V', dir) on all three rows: plane rotations (cs[j], sn[j]) between rows j and j + 1 of A (3 x NCOL, leading dimension lda)
template <int NCOL, bool FORWARD>
MPC_HD void lasr3(const float *cs, const float *sn, float *A, int lda) {
  GELSD_NO_CONTRACT
#pragma unroll
  for (int jj = 0; jj < 2; ++jj) {
    const int j = FORWARD ? jj : 1 - jj;
    const float ct = cs[j], st = sn[j];
    if (ct != 1.f || st != 0.f) {
#pragma unroll
      for (int i = 0; i < NCOL; ++i) {
        const float temp = A[j + 1 + i * lda];
        A[j + 1 + i * lda] = ct * temp - st * A[j + i * lda];
        A[j + i * lda] = st * temp + ct * A[j + i * lda];
      }
    }
  }
}
// the 2 x 2 block (d[R], e[R], d[R + 1]) of SBDSQR: SLASV2, the rotations on rows R, R + 1 of VT and c
template <int R>
MPC_HD void block2(float *d, float *e, float *vt, float *c) {
  float sigmn, sigmx, sinr, cosr, sinl, cosl;
  lasv2(d[R], e[R], d[R + 1], sigmn, sigmx, sinr, cosr, sinl, cosl);
  d[R] = sigmx; e[R] = 0.f; d[R + 1] = sigmn;
  rot<3>(vt + R, 3, vt + R + 1, 3, cosr, sinr);
  rot<1>(c + R, 1, c + R + 1, 1, cosl, sinl);
}
// SBDSQR('U', 3, ncvt = 3, nru = 0, ncc = 1): singular values of the upper bidiagonal (d, e), VT <- rotations applied to the rows of
// the 3 x 3 identity, c <- left rotations applied to c.  Returns 0 (LAPACK info; > 0: no convergence within 6 n^2 sweeps).
// n = 3 leaves the general routine three situations: the unreduced 3 x 3 (ll = 1, m = 3: implicit-shift or zero-shift QR sweeps, chased
// from the larger end), a 2 x 2 block at rows (2, 3) or (1, 2) (SLASV2), and a deflated last value.
MPC_HD int bdsqr3(float *d, float *e, float *vt /* 3 x 3 column-major */, float *c) {
  GELSD_NO_CONTRACT
  const int n = 3, maxitr = 6;
  const float eps = 5.96046448e-08f, unfl = 1.17549435e-38f;
  const float tolmul = 10.f;                                               // max(10, min(100, eps^(-1/8))) with eps^(-1/8) = 2^3
  const float tol = tolmul * eps;
  float smax = fmaxf(fmaxf(fmaxf(fabsf(d[0]), fabsf(d[1])), fabsf(d[2])), fmaxf(fabsf(e[0]), fabsf(e[1])));
  float smin = 0.f;
  float sminoa = fabsf(d[0]);
  if (sminoa != 0.f) {
    float mu = sminoa;
    mu = fabsf(d[1]) * (mu / (mu + fabsf(e[0])));
    sminoa = fminf(sminoa, mu);
    if (sminoa != 0.f) {
      mu = fabsf(d[2]) * (mu / (mu + fabsf(e[1])));
      sminoa = fminf(sminoa, mu);
    }
  }
  sminoa = sminoa / sqrtf((float)n);
  const float thresh = fmaxf(tol * sminoa, (float)maxitr * ((float)n * ((float)n * unfl)));
  const int maxitdivn = maxitr * n;
  int iterdivn = 0, iter = -1, oldll = -1, oldm = -1, m = n, idir = 0;       // m, ll: 1-based as in the Fortran
  for (;;) {
    if (m <= 1) break;
    if (iter >= n) { iter -= n; ++iterdivn; if (iterdivn >= maxitdivn) return 1; }
    if (m == 2) {        // rows 1, 2: a negligible e(1) deflates, otherwise the 2 x 2 block
      if (fabsf(e[0]) <= thresh) { e[0] = 0.f; m = 1; continue; }
      block2<0>(d, e, vt, c);
      m = 0;
      continue;
    }
    // m == 3: look for a split from the bottom
    smax = fabsf(d[2]);
    if (fabsf(e[1]) <= thresh) { e[1] = 0.f; m = 2; continue; }
    smax = fmaxf(smax, fmaxf(fabsf(d[1]), fabsf(e[1])));
    if (fabsf(e[0]) <= thresh) {        // split at the top: the 2 x 2 block of rows 2, 3
      e[0] = 0.f;
      block2<1>(d, e, vt, c);
      m = 1;
      continue;
    }
    smax = fmaxf(smax, fmaxf(fabsf(d[0]), fabsf(e[0])));
    const int ll = 1;
    if (ll > oldm || m < oldll) idir = (fabsf(d[0]) >= fabsf(d[2])) ? 1 : 2;
    if (idir == 1) {
      if (fabsf(e[1]) <= fabsf(tol) * fabsf(d[2])) { e[1] = 0.f; continue; }
      float mu = fabsf(d[0]);
      smin = mu;
      if (fabsf(e[0]) <= tol * mu) { e[0] = 0.f; continue; }
      mu = fabsf(d[1]) * (mu / (mu + fabsf(e[0])));
      smin = fminf(smin, mu);
      if (fabsf(e[1]) <= tol * mu) { e[1] = 0.f; continue; }
      mu = fabsf(d[2]) * (mu / (mu + fabsf(e[1])));
      smin = fminf(smin, mu);
    } else {
      if (fabsf(e[0]) <= fabsf(tol) * fabsf(d[0])) { e[0] = 0.f; continue; }
      float mu = fabsf(d[2]);
      smin = mu;
      if (fabsf(e[1]) <= tol * mu) { e[1] = 0.f; continue; }
      mu = fabsf(d[1]) * (mu / (mu + fabsf(e[1])));
      smin = fminf(smin, mu);
      if (fabsf(e[0]) <= tol * mu) { e[0] = 0.f; continue; }
      mu = fabsf(d[0]) * (mu / (mu + fabsf(e[0])));
      smin = fminf(smin, mu);
    }
    oldll = ll; oldm = m;
    float shift, r;
    if ((float)n * tol * (smin / smax) <= fmaxf(eps, 0.01f * tol)) shift = 0.f;
    else {
      float sll;
      if (idir == 1) { sll = fabsf(d[0]); las2(d[1], e[1], d[2], shift, r); }
      else { sll = fabsf(d[2]); las2(d[0], e[0], d[1], shift, r); }
      if (sll > 0.f) { const float q = shift / sll; if (q * q < eps) shift = 0.f; }
    }
    iter = iter + m - ll;
    float w1[2], w2[2], w3[2], w4[2];      // cs / sn of the right rotations, of the left rotations
    if (shift == 0.f) {
      if (idir == 1) {
        float cs = 1.f, oldcs = 1.f, sn = 0.f, oldsn = 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          lartg(d[i] * cs, e[i], cs, sn, r);
          if (i > 0) e[i - (i > 0 ? 1 : 0)] = oldsn * r;
          lartg(oldcs * r, d[i + 1] * sn, oldcs, oldsn, d[i]);
          w1[i] = cs; w2[i] = sn; w3[i] = oldcs; w4[i] = oldsn;
        }
        const float h = d[2] * cs;
        d[2] = h * oldcs;
        e[1] = h * oldsn;
        lasr3<3, true>(w1, w2, vt, 3);
        lasr3<1, true>(w3, w4, c, 3);
        if (fabsf(e[1]) <= thresh) e[1] = 0.f;
      } else {
        float cs = 1.f, oldcs = 1.f, sn = 0.f, oldsn = 0.f;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          const int i = 2 - k;                  // 0-based row of D(I), I = M .. LL + 1
          lartg(d[i] * cs, e[i - 1], cs, sn, r);
          if (k > 0) e[i] = oldsn * r;
          lartg(oldcs * r, d[i - 1] * sn, oldcs, oldsn, d[i]);
          w1[i - 1] = cs; w2[i - 1] = -sn; w3[i - 1] = oldcs; w4[i - 1] = -oldsn;
        }
        const float h = d[0] * cs;
        d[0] = h * oldcs;
        e[0] = h * oldsn;
        lasr3<3, false>(w3, w4, vt, 3);
        lasr3<1, false>(w1, w2, c, 3);
        if (fabsf(e[0]) <= thresh) e[0] = 0.f;
      }
    } else {
      if (idir == 1) {
        float f = (fabsf(d[0]) - shift) * (sgn(1.f, d[0]) + shift / d[0]);
        float g = e[0];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          float cosr, sinr, cosl, sinl;
          lartg(f, g, cosr, sinr, r);
          if (i > 0) e[i - (i > 0 ? 1 : 0)] = r;
          f = cosr * d[i] + sinr * e[i];
          e[i] = cosr * e[i] - sinr * d[i];
          g = sinr * d[i + 1];
          d[i + 1] = cosr * d[i + 1];
          lartg(f, g, cosl, sinl, r);
          d[i] = r;
          f = cosl * e[i] + sinl * d[i + 1];
          d[i + 1] = cosl * d[i + 1] - sinl * e[i];
          if (i < 1) { g = sinl * e[i + (i < 1 ? 1 : 0)]; e[i + (i < 1 ? 1 : 0)] = cosl * e[i + (i < 1 ? 1 : 0)]; }
          w1[i] = cosr; w2[i] = sinr; w3[i] = cosl; w4[i] = sinl;
        }
        e[1] = f;
        lasr3<3, true>(w1, w2, vt, 3);
        lasr3<1, true>(w3, w4, c, 3);
        if (fabsf(e[1]) <= thresh) e[1] = 0.f;
      } else {
        float f = (fabsf(d[2]) - shift) * (sgn(1.f, d[2]) + shift / d[2]);
        float g = e[1];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          const int i = 2 - k;
          float cosr, sinr, cosl, sinl;
          lartg(f, g, cosr, sinr, r);
          if (k > 0) e[i] = r;
          f = cosr * d[i] + sinr * e[i - 1];
          e[i - 1] = cosr * e[i - 1] - sinr * d[i];
          g = sinr * d[i - 1];
          d[i - 1] = cosr * d[i - 1];
          lartg(f, g, cosl, sinl, r);
          d[i] = r;
          f = cosl * e[i - 1] + sinl * d[i - 1];
          d[i - 1] = cosl * d[i - 1] - sinl * e[i - 1];
          if (k < 1) { g = sinl * e[i - (k < 1 ? 2 : 1)]; e[i - (k < 1 ? 2 : 1)] = cosl * e[i - (k < 1 ? 2 : 1)]; }
          w1[i - 1] = cosr; w2[i - 1] = -sinr; w3[i - 1] = cosl; w4[i - 1] = -sinl;
        }
        e[0] = f;
        if (fabsf(e[0]) <= thresh) e[0] = 0.f;
        lasr3<3, false>(w3, w4, vt, 3);
        lasr3<1, false>(w1, w2, c, 3);
      }
    }
  }
#pragma unroll
  for (int i = 0; i < 3; ++i)
    if (d[i] < 0.f) {
      d[i] = -d[i];
#pragma unroll
      for (int k = 0; k < 3; ++k) vt[i + 3 * k] = -1.f * vt[i + 3 * k];
    }
  // decreasing order (SBDSQR's selection sort: the smallest of the first n + 1 - i goes to place n + 1 - i; ties take the later one)
  auto swap_rows = [&](int p, int q) {      // constant arguments at every call
    const float td = d[p]; d[p] = d[q]; d[q] = td;
#pragma unroll
    for (int k = 0; k < 3; ++k) { const float t = vt[p + 3 * k]; vt[p + 3 * k] = vt[q + 3 * k]; vt[q + 3 * k] = t; }
    const float tc = c[p]; c[p] = c[q]; c[q] = tc;
  };
  {
    int isub = 0;
    float sm = d[0];
    if (d[1] <= sm) { isub = 1; sm = d[1]; }
    if (d[2] <= sm) { isub = 2; sm = d[2]; }
    if (isub == 0) swap_rows(0, 2); else if (isub == 1) swap_rows(1, 2);
    if (!(d[1] <= d[0])) swap_rows(0, 1);
  }
  return 0;
}
// SLASCL('G') for values in the normal range: one rounded quotient, then one product per entry
MPC_HD float lascl_mul(float cfrom, float cto) { return cto / cfrom; }

// x (3) = argmin |A x - 1|, A = 4 x 3 row-major float32 (the foot-contact history).  Returns LAPACK's rank.
MPC_HD int solve_ones(const float *A_rowmajor, float *x, float *dbg = nullptr) {
  GELSD_NO_CONTRACT
  constexpr int lda = 4;
  float a[12], b[4] = {1.f, 1.f, 1.f, 1.f}, tau[3];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) a[i + lda * j] = A_rowmajor[3 * i + j];
  // SGEQR2
  {
    larfg<4>(a[0], a + 1, 1, tau[0]);
    float aii = a[0]; a[0] = 1.f;
    larf_left<4, 2>(a, 1, tau[0], a + lda, lda);
    a[0] = aii;
    larfg<3>(a[1 + lda], a + 2 + lda, 1, tau[1]);
    aii = a[1 + lda]; a[1 + lda] = 1.f;
    larf_left<3, 1>(a + 1 + lda, 1, tau[1], a + 1 + 2 * lda, lda);
    a[1 + lda] = aii;
    larfg<2>(a[2 + 2 * lda], a + 3 + 2 * lda, 1, tau[2]);
  }
  // SORM2R('L', 'T'): Q^T b
  {
    float aii = a[0]; a[0] = 1.f;
    larf_left<4, 1>(a, 1, tau[0], b, 4);
    a[0] = aii;
    aii = a[1 + lda]; a[1 + lda] = 1.f;
    larf_left<3, 1>(a + 1 + lda, 1, tau[1], b + 1, 4);
    a[1 + lda] = aii;
    aii = a[2 + 2 * lda]; a[2 + 2 * lda] = 1.f;
    larf_left<2, 1>(a + 2 + 2 * lda, 1, tau[2], b + 2, 4);
    a[2 + 2 * lda] = aii;
  }
  if (dbg) { for (int k = 0; k < 12; ++k) dbg[k] = a[k]; for (int k = 0; k < 3; ++k) dbg[12 + k] = tau[k]; for (int k = 0; k < 4; ++k) dbg[15 + k] = b[k]; }
  a[1] = a[2] = a[2 + lda] = 0.f;                                  // SLASET below the diagonal of R
  // SGEBD2 on the 3 x 3 R (lda = 4)
  float d[3], e[2], tauq[3], taup[3];
  {
    // i = 1: the column below R(1,1) is zero -> tauq(1) = 0; the row reflector on R(1, 2:3)
    larfg<3>(a[0], a + 1, 1, tauq[0]);
    d[0] = a[0]; a[0] = 1.f;
    larf_left<3, 2>(a, 1, tauq[0], a + lda, lda);
    a[0] = d[0];
    larfg<2>(a[lda], a + 2 * lda, lda, taup[0]);
    e[0] = a[lda]; a[lda] = 1.f;
    larf_right<2, 2>(a + lda, lda, taup[0], a + 1 + lda, lda);
    a[lda] = e[0];
    // i = 2
    larfg<2>(a[1 + lda], a + 2 + lda, 1, tauq[1]);
    d[1] = a[1 + lda]; a[1 + lda] = 1.f;
    larf_left<2, 1>(a + 1 + lda, 1, tauq[1], a + 1 + 2 * lda, lda);
    a[1 + lda] = d[1];
    taup[1] = 0.f;                                                  // SLARFG on one entry
    e[1] = a[1 + 2 * lda];
    // i = 3
    tauq[2] = 0.f; taup[2] = 0.f;
    d[2] = a[2 + 2 * lda];
  }
  // SORMBR('Q', 'L', 'T') = SORM2R with the tauq reflectors (only tauq(2) can be non-zero; tauq(1) as SLARFG left it)
  {
    float aii = a[0]; a[0] = 1.f;
    larf_left<3, 1>(a, 1, tauq[0], b, 4);
    a[0] = aii;
    aii = a[1 + lda]; a[1 + lda] = 1.f;
    larf_left<2, 1>(a + 1 + lda, 1, tauq[1], b + 1, 4);
    a[1 + lda] = aii;
  }
  if (dbg) { for (int k = 0; k < 12; ++k) dbg[20 + k] = a[k]; for (int k = 0; k < 3; ++k) { dbg[32 + k] = d[k]; dbg[37 + k] = tauq[k]; dbg[40 + k] = taup[k]; dbg[43 + k] = b[k]; } dbg[35] = e[0]; dbg[36] = e[1]; }
  // SLALSD('U', n = 3 <= smlsiz)
  const float rcnd = 1.1920929e-07f;                               // scipy passes cond = finfo(float32).eps
  const float orgnrm = fmaxf(fmaxf(fmaxf(fabsf(d[0]), fabsf(d[1])), fabsf(d[2])), fmaxf(fabsf(e[0]), fabsf(e[1])));
  int rank = 0;
  if (orgnrm == 0.f) { x[0] = x[1] = x[2] = 0.f; return 0; }
  const float unit = lascl_mul(orgnrm, 1.f);
  if (unit != 1.f) { d[0] = d[0] * unit; d[1] = d[1] * unit; d[2] = d[2] * unit; e[0] = e[0] * unit; e[1] = e[1] * unit; }
  float vt[9] = {1.f, 0.f, 0.f, 0.f, 1.f, 0.f, 0.f, 0.f, 1.f};
  if (bdsqr3(d, e, vt, b) != 0) { x[0] = x[1] = x[2] = 0.f; return -1; }
  {    // SLASDQ re-sorts what SBDSQR left in decreasing order into INCREASING order (selection sort, strict comparisons)
    auto swap_rows = [&](int p, int q) {
      const float td = d[p]; d[p] = d[q]; d[q] = td;
#pragma unroll
      for (int k = 0; k < 3; ++k) { const float t = vt[p + 3 * k]; vt[p + 3 * k] = vt[q + 3 * k]; vt[q + 3 * k] = t; }
      const float tc = b[p]; b[p] = b[q]; b[q] = tc;
    };
    int isub = 0;
    float sm = d[0];
    if (d[1] < sm) { isub = 1; sm = d[1]; }
    if (d[2] < sm) { isub = 2; sm = d[2]; }
    if (isub == 1) swap_rows(0, 1); else if (isub == 2) swap_rows(0, 2);
    if (d[2] < d[1]) swap_rows(1, 2);
  }
  const float dmax = fmaxf(fmaxf(fabsf(d[0]), fabsf(d[1])), fabsf(d[2]));
  const float tol = rcnd * dmax;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    if (d[i] <= tol) b[i] = 0.f;
    else { const float mul = lascl_mul(d[i], 1.f); if (mul != 1.f) b[i] = b[i] * mul; ++rank; }
  }
  float y[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) y[i] = fmaf(vt[2 + 3 * i], b[2], fmaf(vt[1 + 3 * i], b[1], vt[0 + 3 * i] * b[0]));      // SGEMM('T', 'N', 3, 1, 3)
  if (unit != 1.f) { y[0] = y[0] * unit; y[1] = y[1] * unit; y[2] = y[2] * unit; }
  b[0] = y[0]; b[1] = y[1]; b[2] = y[2];
  if (dbg) for (int k = 0; k < 3; ++k) dbg[46 + k] = b[k];
  // SORMBR('P', 'L', 'N') = SORML2('L', 'T', 2, 1, 2) on b(2:3): reflectors taup(2) (= 0), then taup(1)
  {
    const float aii = a[lda];
    a[lda] = 1.f;
    larf_left<2, 1>(a + lda, lda, taup[0], b + 1, 4);
    a[lda] = aii;
  }
  x[0] = b[0]; x[1] = b[1]; x[2] = b[2];
  return rank;
}

}  // namespace gelsd43
}  // namespace mpc
