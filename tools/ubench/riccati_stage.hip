// Microbenchmark behind profiles/r06_ab_riccati_*.txt: the wrench-space core system of the solve kernel,
//     M y = g,   M = I + L^T (c Theta) L   (6h x 6h; csrc/mpc_wrench.h),
// solved STAGE-WISE by ONE wavefront per robot instead of through the explicit inverse in 6 x 6 register tiles.
// Theta is the Hessian of a 12-state double integrator driven by the per-step wrenches u_k = L_k y_k:
//     v_{k+1} = v_k + u_k,  pi_{k+1} = pi_k + v_k + u_k / 2,   cost sum_k  pi_k^T th1 pi_k + v_k^T diag(th2) v_k   (k = 1 .. h)
// (Theta_{jj'} = s2 th1 + n diag(th2), mpc_wrench.h th_s2 / th_nn), so M y = g is an LQ problem: a backward Riccati recursion of h
// steps (12 x 12 cost-to-go, 6 x 6 pivot block per step) factors it, a backward + a forward sweep of h steps each applies M^-1.
//   hipcc --offload-arch=gfx950 -O3 riccati_stage.hip -o riccati_stage && ./riccati_stage
// Prints, per horizon and waves per SIMD: shader cycles per factorisation and per application, max error against a dense solve.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <int H>
struct Lds {
  double L[H][36];      // L_k, row-major, lower triangular
  double K[H][72];      // K_k [6][12]
  double Hi[H][36];     // (I + Bt^T P Bt)^-1
  double Phi[H][144];   // A - Bt K_k, row-major
  double P[2][144];
  double W[72], Hm[36], Kt[72];
  double g[H][6], cv[H][12], pall[H + 1][12], r[H][6], d[H][6], e[H][12], xall[H + 1][12];
  double th1[36], th2[6];
};

__device__ __forceinline__ double rcp(double d) {
  double r = __builtin_amdgcn_rcp(d);
  r = r * (2.0 - d * r);
  r = r * (2.0 - d * r);
  return r;
}

template <int H>
__device__ void factor(Lds<H> &s, double c, int l, long long *ph4) {
  // P_h = c Qs
  for (int e = l; e < 144; e += 64) {
    const int a = e / 12, b = e % 12;
    s.P[0][e] = a < 6 && b < 6 ? c * s.th1[6 * a + b] : (a == b ? c * s.th2[a - 6] : 0.0);
  }
  __syncthreads();
  int cur = 0;
  for (int k = H - 1; k >= 0; --k) {
    const double *P = s.P[cur], *L = s.L[k];
    double *Pn = s.P[cur ^ 1];
    const long long q0 = __builtin_readcyclecounter();
    {   // phase 1: W = P Bt,  Bt = [L / 2; L]
      const int a = l & 15, jj = l >> 4;
      if (a < 12) {
        double cm[6];
#pragma unroll
        for (int m = 0; m < 6; ++m) cm[m] = 0.5 * P[12 * a + m] + P[12 * a + 6 + m];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const int j = jj + 4 * t;
          if (j < 6) {
            double acc = 0;
#pragma unroll
            for (int m = 0; m < 6; ++m) acc += m >= j ? cm[m] * L[6 * m + j] : 0.0;
            s.W[6 * a + j] = acc;
          }
        }
      }
    }
    __syncthreads();
    const long long q1 = __builtin_readcyclecounter();
    if (l < 36) {   // phase 2: Hm = I + Bt^T W
      const int i = l / 6, j = l % 6;
      double acc = i == j ? 1.0 : 0.0;
#pragma unroll
      for (int m = 0; m < 6; ++m) acc += m >= i ? L[6 * m + i] * (0.5 * s.W[6 * m + j] + s.W[6 * (6 + m) + j]) : 0.0;
      s.Hm[l] = acc;
    }
    __syncthreads();
    const long long q2 = __builtin_readcyclecounter();
    if (l < 18) {   // phase 3: Hm^-1 (columns 0..5) and K = Hm^-1 Y (columns 6..17), L D L^T per lane
      double h[21], dd[6], di[6], li[21];
      auto pk = [](int r, int cc) { return r * (r + 1) / 2 + cc; };
#pragma unroll
      for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = 0; j <= i; ++j) h[pk(i, j)] = s.Hm[6 * i + j];
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        double dj = h[pk(j, j)];
#pragma unroll
        for (int q = 0; q < j; ++q) dj -= li[pk(j, q)] * li[pk(j, q)] * dd[q];
        const double inv = rcp(dj);
        dd[j] = dj; di[j] = inv;
#pragma unroll
        for (int i = j + 1; i < 6; ++i) {
          double v = h[pk(i, j)];
#pragma unroll
          for (int q = 0; q < j; ++q) v -= li[pk(i, q)] * li[pk(j, q)] * dd[q];
          li[pk(i, j)] = v * inv;
        }
      }
      double x[6];
      const int jc = l - 6;
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        if (l < 6) x[i] = i == l ? 1.0 : 0.0;
        else x[i] = jc < 6 ? s.W[6 * jc + i] : s.W[6 * (jc - 6) + i] + s.W[6 * jc + i];
      }
#pragma unroll
      for (int i = 1; i < 6; ++i)
#pragma unroll
        for (int q = 0; q < i; ++q) x[i] -= li[pk(i, q)] * x[q];
#pragma unroll
      for (int i = 0; i < 6; ++i) x[i] *= di[i];
#pragma unroll
      for (int i = 4; i >= 0; --i)
#pragma unroll
        for (int q = i + 1; q < 6; ++q) x[i] -= li[pk(q, i)] * x[q];
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        if (l < 6) s.Hi[k][6 * i + l] = x[i];
        else s.K[k][12 * i + jc] = x[i];
      }
    }
    __syncthreads();
    const long long q3 = __builtin_readcyclecounter();
    // phase 4: P_k = c Qs + A^T P A - Y^T K;  Phi = A - Bt K
    for (int e = l; e < 144; e += 64) {
      const int a = e / 12, b = e % 12, ia = a % 6, ib = b % 6;
      double v = a < 6 && b < 6 ? c * s.th1[6 * a + b] : (a == b ? c * s.th2[a - 6] : 0.0);
      v += P[12 * ia + ib];
      if (b >= 6) v += P[12 * ia + 6 + ib];
      if (a >= 6) v += P[12 * (6 + ia) + ib];
      if (a >= 6 && b >= 6) v += P[12 * (6 + ia) + 6 + ib];
      double lk = 0;
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        const double y = a < 6 ? s.W[6 * a + i] : s.W[6 * (a - 6) + i] + s.W[6 * a + i];
        v -= y * s.K[k][12 * i + b];
        lk += i <= ia ? L[6 * ia + i] * s.K[k][12 * i + b] : 0.0;
      }
      Pn[e] = v;
      const double A = (a == b ? 1.0 : 0.0) + (a < 6 && b == a + 6 ? 1.0 : 0.0);
      s.Phi[k][e] = A - (a < 6 ? 0.5 * lk : lk);
    }
    __syncthreads();
    const long long q4 = __builtin_readcyclecounter();
    ph4[0] += q1 - q0; ph4[1] += q2 - q1; ph4[2] += q3 - q2; ph4[3] += q4 - q3;
    cur ^= 1;
  }
}

template <int B> __device__ __forceinline__ double bc(double v) {      // lane B of my row of 16 lanes (v_mov_b64_dpp row_newbcast)
  long long x = __builtin_bit_cast(long long, v);
  x = __builtin_amdgcn_update_dpp((long long)0, x, 0x150 + B, 0xF, 0xF, true);
  return __builtin_bit_cast(double, x);
}
__device__ __forceinline__ double dot12(const double *m, double p) {
  double a0 = m[0] * bc<0>(p), a1 = m[1] * bc<1>(p), a2 = m[2] * bc<2>(p);
  a0 = fma(m[3], bc<3>(p), a0); a1 = fma(m[4], bc<4>(p), a1); a2 = fma(m[5], bc<5>(p), a2);
  a0 = fma(m[6], bc<6>(p), a0); a1 = fma(m[7], bc<7>(p), a1); a2 = fma(m[8], bc<8>(p), a2);
  a0 = fma(m[9], bc<9>(p), a0); a1 = fma(m[10], bc<10>(p), a1); a2 = fma(m[11], bc<11>(p), a2);
  return (a0 + a1) + a2;
}
// v2: the 2 h sequential steps exchange the 12-vector through DPP (no LDS round trip), the next step's matrix is prefetched
template <int H>
__device__ void apply2(Lds<H> &s, int l, double *out) {
  const int a12 = l % 12, k12 = l / 12, i6 = l % 6, k6 = l / 6;
  for (int e = l, k = k12; e < 12 * H; e += 64, k += 5) {   // cv_k = K_k^T g_k   (64 = 5 * 12 + 4)
    const int kk = e / 12, a = e - 12 * kk;
    double acc = 0;
#pragma unroll
    for (int i = 0; i < 6; ++i) acc += s.K[kk][12 * i + a] * s.g[kk][i];
    s.cv[kk][a] = acc;
  }
  (void)a12; (void)k12; (void)i6; (void)k6;
  __syncthreads();
  const int la = l < 12 ? l : 0;
  double p = 0, ph[12], cvn;
#pragma unroll
  for (int b = 0; b < 12; ++b) ph[b] = s.Phi[H - 1][12 * b + la];
  cvn = s.cv[H - 1][la];
  for (int k = H - 1; k >= 0; --k) {   // p_k = Phi_k^T p_{k+1} + cv_k
    double pn[12], cn = 0;
    const int kn = k > 0 ? k - 1 : 0;
#pragma unroll
    for (int b = 0; b < 12; ++b) pn[b] = s.Phi[kn][12 * b + la];
    cn = s.cv[kn][la];
    if (l < 12) s.pall[k + 1][l] = p;
    p = cvn + dot12(ph, p);
#pragma unroll
    for (int b = 0; b < 12; ++b) ph[b] = pn[b];
    cvn = cn;
  }
  __syncthreads();
  for (int e = l; e < 6 * H; e += 64) {   // r_k = L_k^T B^T p_{k+1} - g_k
    const int k = e / 6, i = e % 6;
    double acc = -s.g[k][i];
#pragma unroll
    for (int m = 0; m < 6; ++m) acc += m >= i ? s.L[k][6 * m + i] * (0.5 * s.pall[k + 1][m] + s.pall[k + 1][6 + m]) : 0.0;
    s.r[k][i] = acc;
  }
  __syncthreads();
  for (int e = l; e < 6 * H; e += 64) {   // d_k = Hm^-1 r_k
    const int k = e / 6, i = e % 6;
    double acc = 0;
#pragma unroll
    for (int j = 0; j < 6; ++j) acc += s.Hi[k][6 * i + j] * s.r[k][j];
    s.d[k][i] = acc;
  }
  __syncthreads();
  for (int e = l; e < 12 * H; e += 64) {   // e_k = Bt_k d_k
    const int k = e / 12, a = e % 12, ia = a % 6;
    double acc = 0;
#pragma unroll
    for (int j = 0; j < 6; ++j) acc += j <= ia ? s.L[k][6 * ia + j] * s.d[k][j] : 0.0;
    s.e[k][a] = a < 6 ? 0.5 * acc : acc;
  }
  __syncthreads();
  double x = 0, en;
#pragma unroll
  for (int b = 0; b < 12; ++b) ph[b] = s.Phi[0][12 * la + b];
  en = s.e[0][la];
  for (int k = 0; k < H; ++k) {   // x_{k+1} = Phi_k x_k - e_k
    double pn[12], e2;
    const int kn = k + 1 < H ? k + 1 : k;
#pragma unroll
    for (int b = 0; b < 12; ++b) pn[b] = s.Phi[kn][12 * la + b];
    e2 = s.e[kn][la];
    if (l < 12) s.xall[k][l] = x;
    x = dot12(ph, x) - en;
#pragma unroll
    for (int b = 0; b < 12; ++b) ph[b] = pn[b];
    en = e2;
  }
  __syncthreads();
  for (int e = l; e < 6 * H; e += 64) {   // y_k = -K_k x_k - d_k;  out = g - y
    const int k = e / 6, i = e % 6;
    double acc = s.d[k][i];
#pragma unroll
    for (int b = 0; b < 12; ++b) acc += s.K[k][12 * i + b] * s.xall[k][b];
    out[e] = s.g[k][i] + acc;
  }
  __syncthreads();
}

// out <- (I - M^-1) g   (g in s.g)
template <int H>
__device__ void apply(Lds<H> &s, int l, double *out) {
  for (int e = l; e < 12 * H; e += 64) {   // cv_k = K_k^T g_k
    const int k = e / 12, a = e % 12;
    double acc = 0;
#pragma unroll
    for (int i = 0; i < 6; ++i) acc += s.K[k][12 * i + a] * s.g[k][i];
    s.cv[k][a] = acc;
  }
  __syncthreads();
  double p = 0;
  for (int k = H - 1; k >= 0; --k) {   // p_k = Phi_k^T p_{k+1} + cv_k
    if (l < 12) s.pall[k + 1][l] = p;
    __syncthreads();
    if (l < 12) {
      double acc = s.cv[k][l];
#pragma unroll
      for (int b = 0; b < 12; ++b) acc += s.Phi[k][12 * b + l] * s.pall[k + 1][b];
      p = acc;
    }
  }
  __syncthreads();
  for (int e = l; e < 6 * H; e += 64) {   // r_k = L_k^T B^T p_{k+1} - g_k
    const int k = e / 6, i = e % 6;
    double acc = -s.g[k][i];
#pragma unroll
    for (int m = 0; m < 6; ++m) acc += m >= i ? s.L[k][6 * m + i] * (0.5 * s.pall[k + 1][m] + s.pall[k + 1][6 + m]) : 0.0;
    s.r[k][i] = acc;
  }
  __syncthreads();
  for (int e = l; e < 6 * H; e += 64) {   // d_k = Hm^-1 r_k
    const int k = e / 6, i = e % 6;
    double acc = 0;
#pragma unroll
    for (int j = 0; j < 6; ++j) acc += s.Hi[k][6 * i + j] * s.r[k][j];
    s.d[k][i] = acc;
  }
  __syncthreads();
  for (int e = l; e < 12 * H; e += 64) {   // e_k = Bt_k d_k
    const int k = e / 12, a = e % 12, ia = a % 6;
    double acc = 0;
#pragma unroll
    for (int j = 0; j < 6; ++j) acc += j <= ia ? s.L[k][6 * ia + j] * s.d[k][j] : 0.0;
    s.e[k][a] = a < 6 ? 0.5 * acc : acc;
  }
  __syncthreads();
  double x = 0;
  for (int k = 0; k < H; ++k) {   // x_{k+1} = Phi_k x_k - e_k
    if (l < 12) s.xall[k][l] = x;
    __syncthreads();
    if (l < 12) {
      double acc = -s.e[k][l];
#pragma unroll
      for (int b = 0; b < 12; ++b) acc += s.Phi[k][12 * l + b] * s.xall[k][b];
      x = acc;
    }
  }
  __syncthreads();
  for (int e = l; e < 6 * H; e += 64) {   // y_k = -K_k x_k - d_k;  out = g - y
    const int k = e / 6, i = e % 6;
    double acc = s.d[k][i];
#pragma unroll
    for (int b = 0; b < 12; ++b) acc += s.K[k][12 * i + b] * s.xall[k][b];
    out[e] = s.g[k][i] + acc;
  }
  __syncthreads();
}


// ---------------------------------------------------------------------------------------------------------------------------------
// v3: the same algorithm laid out for one wavefront that runs alone on its SIMD (every fp64 instruction ~6-8 cycles, every LDS hand-over ~130):
//   * P+ = P - W Hm^-1 W^T is what is kept (lower triangle computed, both halves stored); the next step reads Qs + A^T P+ A row by row (adds only);
//   * lane maps are static (no integer division inside the loops), dot products of triangular factors skip their structural zeros;
//   * the 2 h sequential steps of an application pass the 12-vector through v_fmac_f64_dpp row_newbcast (one instruction per term, no LDS round trip),
//     the next step's matrix row is prefetched.
template <int B> __device__ __forceinline__ void fmac_bc(double &acc, double p, double m) {      // acc += (lane B of my row).p * m
  asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(p), "v"(m), "n"(B));
}
__device__ __forceinline__ double dot12_dpp(const double *m, double p) {
  double a0 = 0, a1 = 0, a2 = 0;
  asm volatile("s_nop 1" ::: "memory");      // (VALU write of p -> DPP read: two wait states; inline asm is outside the compiler's hazard recogniser)
  fmac_bc<0>(a0, p, m[0]); fmac_bc<1>(a1, p, m[1]); fmac_bc<2>(a2, p, m[2]);
  fmac_bc<3>(a0, p, m[3]); fmac_bc<4>(a1, p, m[4]); fmac_bc<5>(a2, p, m[5]);
  fmac_bc<6>(a0, p, m[6]); fmac_bc<7>(a1, p, m[7]); fmac_bc<8>(a2, p, m[8]);
  fmac_bc<9>(a0, p, m[9]); fmac_bc<10>(a1, p, m[10]); fmac_bc<11>(a2, p, m[11]);
  return (a0 + a1) + a2;
}

template <int H>
__device__ void factor3(Lds<H> &s, double c, int l, long long *ph4) {
  // static lane roles
  const int a1 = l & 15, jg = l >> 4;                         // phase 1: row a1 (< 12) of W, columns jg and jg + 4
  const int i2 = l / 6, j2 = l - 6 * i2;                      // phase 2: entry (i2, j2) of Hm (l < 36)
  int ra[2], rb[2];                                           // phase 4: my one or two entries (a >= b) of P+
  for (int t = 0; t < 2; ++t) {
    const int e = l + 64 * t;
    int a = 0;
    while ((a + 1) * (a + 2) / 2 <= e) ++a;
    ra[t] = a; rb[t] = e - a * (a + 1) / 2;
  }
  const bool two = l + 64 < 78;
  const int m5 = l / 12 < 6 ? l / 12 : 5, b5 = l % 12;        // phase 4b: entry (m5, b5) of L K (l < 60), lanes 0..11 also row 5
  // P+ of "step h": zero (P_h = Qs = Qs + A^T 0 A)
  for (int e = l; e < 144; e += 64) s.P[0][e] = 0.0;
  __syncthreads();
  for (int k = H - 1; k >= 0; --k) {
    const double *Pp = s.P[0], *L = s.L[k];
    const long long q0 = __builtin_readcyclecounter();
    if (a1 < 12) {   // phase 1: W = (Qs + A^T P+ A) Bt,  Bt = [L / 2; L]
      const int ia = a1 < 6 ? a1 : a1 - 6;
      double lo[6], hi[6];      // row a1 of Qs + A^T P+ A: columns 0..5 / 6..11
#pragma unroll
      for (int m = 0; m < 6; ++m) {
        const double p00 = Pp[12 * ia + m], p01 = Pp[12 * ia + 6 + m];
        double u = p00, v = p00 + p01;
        if (a1 >= 6) { const double p10 = Pp[12 * (6 + ia) + m], p11 = Pp[12 * (6 + ia) + 6 + m]; u += p10; v += p10 + p11; }
        lo[m] = u + (a1 < 6 ? c * s.th1[6 * ia + m] : 0.0);
        hi[m] = v + (a1 >= 6 && m == ia ? c * s.th2[ia] : 0.0);
      }
      double cm[6];
#pragma unroll
      for (int m = 0; m < 6; ++m) cm[m] = 0.5 * lo[m] + hi[m];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int j = jg + 4 * t;
        if (j < 6) {
          double acc = 0;
#pragma unroll
          for (int m = 0; m < 6; ++m) acc += m >= j ? cm[m] * L[6 * m + j] : 0.0;
          s.W[6 * a1 + j] = acc;
        }
      }
    }
    __syncthreads();
    const long long q1 = __builtin_readcyclecounter();
    if (l < 36) {   // phase 2: Hm = I + Bt^T W
      double acc = i2 == j2 ? 1.0 : 0.0;
#pragma unroll
      for (int m = 0; m < 6; ++m) acc += m >= i2 ? L[6 * m + i2] * (0.5 * s.W[6 * m + j2] + s.W[6 * (6 + m) + j2]) : 0.0;
      s.Hm[l] = acc;
    }
    __syncthreads();
    const long long q2 = __builtin_readcyclecounter();
    if (l < 18) {   // phase 3: columns of Hm^-1 (lanes 0..5) and of Kt = Hm^-1 W^T (lanes 6..17): L D L^T + two triangular solves per lane
      double h[21], dd[6], di[6], li[21];
      auto pk = [](int r, int cc) { return r * (r + 1) / 2 + cc; };
#pragma unroll
      for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = 0; j <= i; ++j) h[pk(i, j)] = s.Hm[6 * i + j];
      double x[6];
      const int jc = l < 6 ? 0 : l - 6;
#pragma unroll
      for (int i = 0; i < 6; ++i) { const double w = s.W[6 * jc + i]; x[i] = l < 6 ? (i == l ? 1.0 : 0.0) : w; }
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        double dj = h[pk(j, j)];
#pragma unroll
        for (int q = 0; q < j; ++q) dj -= (li[pk(j, q)] * dd[q]) * li[pk(j, q)];
        const double inv = rcp(dj);
        dd[j] = dj; di[j] = inv;
#pragma unroll
        for (int i = j + 1; i < 6; ++i) {
          double v = h[pk(i, j)];
#pragma unroll
          for (int q = 0; q < j; ++q) v -= (li[pk(i, q)] * dd[q]) * li[pk(j, q)];
          li[pk(i, j)] = v * inv;
        }
      }
#pragma unroll
      for (int i = 1; i < 6; ++i)
#pragma unroll
        for (int q = 0; q < i; ++q) x[i] -= li[pk(i, q)] * x[q];
#pragma unroll
      for (int i = 0; i < 6; ++i) x[i] *= di[i];
#pragma unroll
      for (int i = 4; i >= 0; --i)
#pragma unroll
        for (int q = i + 1; q < 6; ++q) x[i] -= li[pk(q, i)] * x[q];
      if (l < 6) {
#pragma unroll
        for (int i = 0; i < 6; ++i) s.Hi[k][6 * i + l] = x[i];
      } else {
#pragma unroll
        for (int i = 0; i < 6; ++i) s.Kt[12 * i + jc] = x[i];
      }
    }
    __syncthreads();
    const long long q3 = __builtin_readcyclecounter();
    // phase 4: P+ <- (Qs + A^T P+ A) - W Kt (lower triangle, mirrored);  K = Kt A;  Phi = A - Bt K
    double pn[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      if (t == 0 || two) {
        const int a = ra[t], b = rb[t], ia = a < 6 ? a : a - 6, ib = b < 6 ? b : b - 6;
        double v = Pp[12 * ia + ib];
        if (b >= 6) v += Pp[12 * ia + 6 + ib];
        if (a >= 6) v += Pp[12 * (6 + ia) + ib];
        if (a >= 6 && b >= 6) v += Pp[12 * (6 + ia) + 6 + ib];
        v += a < 6 && b < 6 ? c * s.th1[6 * a + b] : (a == b ? c * s.th2[ia] : 0.0);
#pragma unroll
        for (int i = 0; i < 6; ++i) v -= s.W[6 * a + i] * s.Kt[12 * i + b];
        pn[t] = v;
      }
    }
    if (l < 72) {      // (always true: 64 lanes; lanes 0..11 take row 5 as a second entry below)
      double kk[6];
#pragma unroll
      for (int i = 0; i < 6; ++i) kk[i] = s.Kt[12 * i + (b5 < 6 ? b5 : b5 - 6)] * (b5 < 6 ? 1.0 : 1.0) + (b5 < 6 ? 0.0 : s.Kt[12 * i + b5]);
      // (K[i][b] = Kt[i][b] for b < 6, Kt[i][b - 6] + Kt[i][b] for b >= 6)
      if (b5 < 6) {
#pragma unroll
        for (int i = 0; i < 6; ++i) kk[i] = s.Kt[12 * i + b5];
      }
      if (l < 60) {
        if (m5 == 0) {
#pragma unroll
          for (int i = 0; i < 6; ++i) s.K[k][12 * i + b5] = kk[i];
        }
        double lk = 0;
#pragma unroll
        for (int i = 0; i < 6; ++i) lk += i <= m5 ? L[6 * m5 + i] * kk[i] : 0.0;
        s.Phi[k][12 * m5 + b5] = (m5 == b5 ? 1.0 : 0.0) + (b5 == m5 + 6 ? 1.0 : 0.0) - 0.5 * lk;
        s.Phi[k][12 * (6 + m5) + b5] = (6 + m5 == b5 ? 1.0 : 0.0) - lk;
      }
      if (l < 12) {
        double lk = 0;
#pragma unroll
        for (int i = 0; i < 6; ++i) lk += L[6 * 5 + i] * kk[i];
        s.Phi[k][12 * 5 + b5] = (5 == b5 ? 1.0 : 0.0) + (b5 == 11 ? 1.0 : 0.0) - 0.5 * lk;
        s.Phi[k][12 * 11 + b5] = (11 == b5 ? 1.0 : 0.0) - lk;
      }
    }
    __syncthreads();      // (everybody has read the old P+)
#pragma unroll
    for (int t = 0; t < 2; ++t)
      if (t == 0 || two) { s.P[0][12 * ra[t] + rb[t]] = pn[t]; s.P[0][12 * rb[t] + ra[t]] = pn[t]; }
    __syncthreads();
    const long long q4 = __builtin_readcyclecounter();
    ph4[0] += q1 - q0; ph4[1] += q2 - q1; ph4[2] += q3 - q2; ph4[3] += q4 - q3;
  }
}

template <int H>
__device__ void apply3(Lds<H> &s, int l, double *out, long long *ph4) {
  const long long q0 = __builtin_readcyclecounter();
  {   // cv_k = K_k^T g_k: lane (k0 + 5 t, a), 60 lanes per pass
    const int k0 = l / 12, a = l - 12 * k0;
    if (l < 60)
      for (int k = k0; k < H; k += 5) {
        double acc = 0;
#pragma unroll
        for (int i = 0; i < 6; ++i) acc += s.K[k][12 * i + a] * s.g[k][i];
        s.cv[k][a] = acc;
      }
  }
  __syncthreads();
  const long long q1 = __builtin_readcyclecounter();
  const int la = (l & 15) < 12 ? (l & 15) : 0;      // every row of 16 lanes runs the recursion (rows 1..3 redundantly)
  double p = 0, ph[12], cvn;
#pragma unroll
  for (int b = 0; b < 12; ++b) ph[b] = s.Phi[H - 1][12 * b + la];
  cvn = s.cv[H - 1][la];
#pragma unroll 1
  for (int k = H - 1; k >= 0; --k) {   // p_k = Phi_k^T p_{k+1} + cv_k
    double pn[12], cn;
    const int kn = k > 0 ? k - 1 : 0;
#pragma unroll
    for (int b = 0; b < 12; ++b) pn[b] = s.Phi[kn][12 * b + la];
    cn = s.cv[kn][la];
    if (l < 12) s.pall[k + 1][l] = p;
    p = cvn + dot12_dpp(ph, p);
#pragma unroll
    for (int b = 0; b < 12; ++b) ph[b] = pn[b];
    cvn = cn;
  }
  __syncthreads();
  const long long q2 = __builtin_readcyclecounter();
  {   // d_k = Hm^-1 (L_k^T B^T p_{k+1} - g_k): lane (k, i) forms the whole r_k (triangular, 21 terms) and its own row of Hm^-1
    const int k0 = l / 6, i = l - 6 * k0;
    if (l < 60)
      for (int k = k0; k < H; k += 10) {
        double t6[6], r[6];
#pragma unroll
        for (int m = 0; m < 6; ++m) t6[m] = 0.5 * s.pall[k + 1][m] + s.pall[k + 1][6 + m];
#pragma unroll
        for (int j = 0; j < 6; ++j) {
          double acc = -s.g[k][j];
#pragma unroll
          for (int m = j; m < 6; ++m) acc += s.L[k][6 * m + j] * t6[m];
          r[j] = acc;
        }
        double acc = 0;
#pragma unroll
        for (int j = 0; j < 6; ++j) acc += s.Hi[k][6 * i + j] * r[j];
        s.d[k][i] = acc;
      }
  }
  __syncthreads();
  {   // e_k = Bt_k d_k
    const int k0 = l / 12, a = l - 12 * k0, ia = a < 6 ? a : a - 6;
    if (l < 60)
      for (int k = k0; k < H; k += 5) {
        double acc = 0;
#pragma unroll
        for (int j = 0; j < 6; ++j) acc += j <= ia ? s.L[k][6 * ia + j] * s.d[k][j] : 0.0;
        s.e[k][a] = a < 6 ? 0.5 * acc : acc;
      }
  }
  __syncthreads();
  const long long q3 = __builtin_readcyclecounter();
  double x = 0, en;
#pragma unroll
  for (int b = 0; b < 12; ++b) ph[b] = s.Phi[0][12 * la + b];
  en = s.e[0][la];
#pragma unroll 1
  for (int k = 0; k < H; ++k) {   // x_{k+1} = Phi_k x_k - e_k
    double pn[12], e2;
    const int kn = k + 1 < H ? k + 1 : k;
#pragma unroll
    for (int b = 0; b < 12; ++b) pn[b] = s.Phi[kn][12 * la + b];
    e2 = s.e[kn][la];
    if (l < 12) s.xall[k][l] = x;
    x = dot12_dpp(ph, x) - en;
#pragma unroll
    for (int b = 0; b < 12; ++b) ph[b] = pn[b];
    en = e2;
  }
  __syncthreads();
  const long long q4 = __builtin_readcyclecounter();
  {   // y_k = -K_k x_k - d_k;  out = g - y
    const int k0 = l / 6, i = l - 6 * k0;
    if (l < 60)
      for (int k = k0; k < H; k += 10) {
        double acc = s.d[k][i];
#pragma unroll
        for (int b = 0; b < 12; ++b) acc += s.K[k][12 * i + b] * s.xall[k][b];
        out[6 * k + i] = s.g[k][i] + acc;
      }
  }
  __syncthreads();
  const long long q5 = __builtin_readcyclecounter();
  ph4[0] += (q1 - q0) + (q3 - q2) + (q5 - q4); ph4[1] += q2 - q1; ph4[2] += q4 - q3;
}

template <int H, int V>
__global__ __launch_bounds__(64) void riccati_kernel(const double *Lin, const double *th, const double *gin, double c, int napply, double *out, long long *cyc) {
  extern __shared__ __attribute__((aligned(16))) unsigned char raw[];
  Lds<H> &s = *reinterpret_cast<Lds<H> *>(raw);
  const int l = threadIdx.x, rb = blockIdx.x;
  for (int e = l; e < 36 * H; e += 64) (&s.L[0][0])[e] = Lin[(size_t)rb * 36 * H + e];
  for (int e = l; e < 36; e += 64) s.th1[e] = th[e];
  if (l < 6) s.th2[l] = th[36 + l];
  for (int e = l; e < 6 * H; e += 64) (&s.g[0][0])[e] = gin[(size_t)rb * 6 * H + e];
  __syncthreads();
  const long long t0 = __builtin_readcyclecounter();
  long long ph4[4] = {0, 0, 0, 0};
  long long pa[4] = {0, 0, 0, 0};
  if (V == 3) factor3<H>(s, c, l, ph4); else factor<H>(s, c, l, ph4);
  if (l == 0 && rb == 0 && napply > 1) printf("   factor phases (cycles per step): W=P Bt %lld, Hm %lld, LDLt+solves %lld, P/Phi update %lld\n", ph4[0] / H, ph4[1] / H, ph4[2] / H, ph4[3] / H);
  const long long t1 = __builtin_readcyclecounter();
  double *o = out + (size_t)rb * 6 * H;
  for (int it = 0; it < napply; ++it) {
    if (V == 3) apply3<H>(s, l, o, pa); else if (V == 2) apply2<H>(s, l, o); else apply<H>(s, l, o);
    if (it + 1 < napply)
      for (int e = l; e < 6 * H; e += 64) (&s.g[0][0])[e] = gin[(size_t)rb * 6 * H + e] + 1e-3 * o[e];   // the next right-hand side depends on this result
    __syncthreads();
  }
  const long long t2 = __builtin_readcyclecounter();
  if (l == 0 && rb == 0 && napply > 1 && V == 3) printf("   apply phases (cycles per application): parallel phases %lld, backward recursion %lld, forward recursion %lld\n", pa[0] / napply, pa[1] / napply, pa[2] / napply);
  if (l == 0) { cyc[2 * rb] = t1 - t0; cyc[2 * rb + 1] = t2 - t1; }
}

template <int H, int V = 1>
void run(int waves_per_simd) {
  const int n = 256 * 4 * waves_per_simd, NW = 6 * H, napply = 50;
  std::vector<double> L((size_t)n * 36 * H, 0.0), th(42), g((size_t)n * NW), out((size_t)n * NW);
  srand(7 + H);
  auto rnd = [] { return rand() / (double)RAND_MAX - 0.5; };
  double a1[36];
  for (double &v : a1) v = rnd();
  for (int i = 0; i < 6; ++i)
    for (int j = 0; j < 6; ++j) {
      double v = 0;
      for (int q = 0; q < 6; ++q) v += a1[6 * i + q] * a1[6 * j + q];
      th[6 * i + j] = 0.02 * v + (i == j ? 0.01 : 0.0);
    }
  for (int i = 0; i < 6; ++i) th[36 + i] = 0.5 + rnd();
  for (int r = 0; r < n; ++r)
    for (int k = 0; k < H; ++k)
      for (int i = 0; i < 6; ++i)
        for (int j = 0; j <= i; ++j) L[((size_t)r * H + k) * 36 + 6 * i + j] = i == j ? 1.0 + rnd() : 0.6 * rnd();
  for (double &v : g) v = rnd();
  const double c = 0.7;
  double *dL, *dth, *dg, *dout;
  long long *dc;
  hipMalloc(&dL, L.size() * 8); hipMalloc(&dth, th.size() * 8); hipMalloc(&dg, g.size() * 8); hipMalloc(&dout, out.size() * 8); hipMalloc(&dc, sizeof(long long) * 2 * n);
  hipMemcpy(dL, L.data(), L.size() * 8, hipMemcpyHostToDevice); hipMemcpy(dth, th.data(), th.size() * 8, hipMemcpyHostToDevice); hipMemcpy(dg, g.data(), g.size() * 8, hipMemcpyHostToDevice);
  hipFuncSetAttribute(reinterpret_cast<const void *>(riccati_kernel<H, V>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(Lds<H>));
  // (1) correctness: one application
  riccati_kernel<H, V><<<n, 64, sizeof(Lds<H>)>>>(dL, dth, dg, c, 1, dout, dc);
  hipDeviceSynchronize();
  hipMemcpy(out.data(), dout, out.size() * 8, hipMemcpyDeviceToHost);
  double worst = 0;
  for (int r = 0; r < 3; ++r) {   // dense reference: M = I + L^T (c Theta) L, Gaussian elimination
    std::vector<double> Th((size_t)NW * NW), M((size_t)NW * NW, 0.0), y(NW);
    for (int ti = 0; ti < H; ++ti)
      for (int tj = 0; tj < H; ++tj) {
        const int hi = ti > tj ? ti : tj, dd = abs(ti - tj);
        const double m = H - hi, s2 = m * (4 * m * m - 1) / 12 + dd * m * m / 2;
        for (int a = 0; a < 6; ++a)
          for (int b = 0; b < 6; ++b) Th[(size_t)(6 * ti + a) * NW + 6 * tj + b] = c * (s2 * th[6 * a + b] + (a == b ? m * th[36 + a] : 0.0));
      }
    const double *Lr = L.data() + (size_t)r * 36 * H;
    std::vector<double> TL((size_t)NW * NW, 0.0);
    for (int i = 0; i < NW; ++i)
      for (int tj = 0; tj < H; ++tj)
        for (int b = 0; b < 6; ++b) {
          double v = 0;
          for (int q = 0; q < 6; ++q) v += Th[(size_t)i * NW + 6 * tj + q] * Lr[tj * 36 + 6 * q + b];
          TL[(size_t)i * NW + 6 * tj + b] = v;
        }
    for (int ti = 0; ti < H; ++ti)
      for (int a = 0; a < 6; ++a)
        for (int j = 0; j < NW; ++j) {
          double v = 0;
          for (int q = 0; q < 6; ++q) v += Lr[ti * 36 + 6 * q + a] * TL[(size_t)(6 * ti + q) * NW + j];
          M[(size_t)(6 * ti + a) * NW + j] = v + (6 * ti + a == j ? 1.0 : 0.0);
        }
    for (int i = 0; i < NW; ++i) y[i] = g[(size_t)r * NW + i];
    for (int p = 0; p < NW; ++p) {
      const double inv = 1.0 / M[(size_t)p * NW + p];
      for (int i = p + 1; i < NW; ++i) {
        const double f = M[(size_t)i * NW + p] * inv;
        for (int j = p; j < NW; ++j) M[(size_t)i * NW + j] -= f * M[(size_t)p * NW + j];
        y[i] -= f * y[p];
      }
    }
    for (int i = NW - 1; i >= 0; --i) {
      double v = y[i];
      for (int j = i + 1; j < NW; ++j) v -= M[(size_t)i * NW + j] * y[j];
      y[i] = v / M[(size_t)i * NW + i];
    }
    for (int i = 0; i < NW; ++i) worst = fmax(worst, fabs(out[(size_t)r * NW + i] - (g[(size_t)r * NW + i] - y[i])));
  }
  // (2) timing
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  riccati_kernel<H, V><<<n, 64, sizeof(Lds<H>)>>>(dL, dth, dg, c, napply, dout, dc);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  std::vector<long long> cy(2 * n);
  hipMemcpy(cy.data(), dc, sizeof(long long) * 2 * n, hipMemcpyDeviceToHost);
  double f = 0, a = 0;
  for (int r = 0; r < n; ++r) { f += cy[2 * r]; a += cy[2 * r + 1]; }
  printf("v%d h=%2d  %d robots (%d waves/SIMD, LDS %zu B/robot): factor %.0f cycles, apply %.0f cycles (%d chained), kernel %.3f ms; max |err| vs dense solve %.2e\n", V, H, n, waves_per_simd, sizeof(Lds<H>),
         f / n, a / n / napply, napply, ms, worst);
  hipFree(dL); hipFree(dth); hipFree(dg); hipFree(dout); hipFree(dc);
}

int main() {
  run<16>(1); run<20>(1);
  run<16, 2>(1); run<20, 2>(1);
  run<10, 3>(1); run<16, 3>(1); run<20, 3>(1); run<16, 3>(2); run<20, 3>(2);
  return 0;
}
