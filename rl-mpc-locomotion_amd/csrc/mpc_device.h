// mpc_device.h -- how the phase sequences of mpc_core.h / mpc_wrench.h execute on the device: Exec::par = a phase + the hand-over
// between thread roles (s_barrier, or nothing but program order where the workgroup is one wavefront), the quad / wavefront
// reductions as DPP register exchanges.  Device code only (the host emulation has its own Exec, tests/emu/emu.cpp).
#pragma once
#include <hip/hip_runtime.h>

namespace mpc {

constexpr int kWaitVm0 = 0x0F70;   // s_waitcnt vmcnt(0) (gfx9 encoding: vmcnt = simm16[3:0] | [15:14], expcnt [6:4] = 7, lgkmcnt [11:8] = 15: not waited for)

// A double moved between the lanes of a quad (lanes 4 q .. 4 q + 3) with DPP quad permutes: two v_mov_b32_dpp, no LDS.
template <int CTRL>
__device__ __forceinline__ double quad_perm(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_mov_dpp(lo, CTRL, 0xF, 0xF, true);
  hi = __builtin_amdgcn_mov_dpp(hi, CTRL, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}
constexpr int quad_ctrl(int a, int b, int c, int d) { return a | (b << 2) | (c << 4) | (d << 6); }
__device__ __forceinline__ double read_lane(double v, int lane) {   // a lane's value as a wavefront-uniform scalar
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane), hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
  return __hiloint2double(hi, lo);
}

// WAVE: the workgroup is a single wavefront (the h = 10 solve kernel).  Its LDS instructions execute in program order, so a
// phase boundary needs neither s_barrier nor a wait for the stores to land (the loads of the next phase queue up behind them):
// only the compiler has to keep the order (wavefront-scope fence).
template <class TH, bool WAVE = false>
struct DeviceExec {
  TH &th;
  __device__ __forceinline__ TH &first() { return th; }   // (after a workgroup-wide reduction every thread holds the same value)
  template <class F>
  __device__ __forceinline__ void par(F &&f) {
    f(th);
    if constexpr (WAVE) {
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
    } else __syncthreads();
  }
  // a phase that hands nothing over through LDS (its results stay in registers or go to the quad operations below)
  template <class F>
  __device__ __forceinline__ void seq(F &&f) { f(th); }
  // acc(th)[0 .. N) <- the sum over the four lanes of the quad, the same bits in every lane: (l0 + l1) + (l2 + l3)
  template <int N, class A>
  __device__ __forceinline__ void quad_allsum(A &&acc) {
    double *v = acc(th);
#pragma unroll
    for (int i = 0; i < N; ++i) {
      const double a = v[i] + quad_perm<quad_ctrl(1, 0, 3, 2)>(v[i]);
      v[i] = a + quad_perm<quad_ctrl(2, 3, 0, 1)>(a);
    }
  }
  // Reduce-scatter of six per-lane values over the quad: dst(th)[0] <- the quad's sum of src[j] in lane j, dst(th)[1] <- the sum of
  // src[4 + (j & 1)], each with the association of quad_allsum, (l_j + l_j^1) + (l_j^2 + l_j^3).  A lane hands its partner what the
  // partner keeps: three exchanges with lane j ^ 1, two with lane j ^ 2 (the all-sum of all six takes twelve, and a select after it).
  template <class S, class D>
  __device__ __forceinline__ void quad_scatter6(S &&src, D &&dst) {
    const double *w = src(th);
    double *g = dst(th);
    const bool odd = threadIdx.x & 1, hi = threadIdx.x & 2;
    double a[3];
#pragma unroll
    for (int p = 0; p < 3; ++p) {      // lanes 0, 2 keep the components 0, 2, 4 of their pair; lanes 1, 3 keep 1, 3, 5
      const double keep = odd ? w[2 * p + 1] : w[2 * p], give = odd ? w[2 * p] : w[2 * p + 1];
      a[p] = keep + quad_perm<quad_ctrl(1, 0, 3, 2)>(give);
    }
    const double keep = hi ? a[1] : a[0], give = hi ? a[0] : a[1];     // lanes 0, 1 end with component 0 / 1, lanes 2, 3 with 2 / 3
    g[0] = keep + quad_perm<quad_ctrl(2, 3, 0, 1)>(give);
    g[1] = a[2] + quad_perm<quad_ctrl(2, 3, 0, 1)>(a[2]);
  }
  // acc(th)[0] <- the sum, acc(th)[1] <- the maximum (of non-negative values) over the 64 lanes of the wavefront, the same bits in
  // every lane: four DPP steps leave every lane with its row's (16 lanes) result, v_readlane fetches the four rows
  template <class A>
  __device__ __forceinline__ void wave_sum_max(A &&acc) {
    double *v = acc(th);
    double a = v[0], m = v[1];
    a += quad_perm<quad_ctrl(1, 0, 3, 2)>(a);  m = fmax(m, quad_perm<quad_ctrl(1, 0, 3, 2)>(m));
    a += quad_perm<quad_ctrl(2, 3, 0, 1)>(a);  m = fmax(m, quad_perm<quad_ctrl(2, 3, 0, 1)>(m));
    a += quad_perm<0x141>(a);                  m = fmax(m, quad_perm<0x141>(m));     // row_half_mirror
    a += quad_perm<0x140>(a);                  m = fmax(m, quad_perm<0x140>(m));     // row_mirror
    v[0] = (read_lane(a, 0) + read_lane(a, 16)) + (read_lane(a, 32) + read_lane(a, 48));
    v[1] = fmax(fmax(read_lane(m, 0), read_lane(m, 16)), fmax(read_lane(m, 32), read_lane(m, 48)));
  }
  // val(th)[0] <- the maximum over the workgroup's threads, idx(th) <- the lowest thread holding it (the same in every thread).
  // One wavefront: DPP row reductions + readlane + a ballot; several: LDS scratch (>= blockDim.x doubles) and two barriers.
  template <class V, class I>
  __device__ __forceinline__ void wg_argmax(V &&val, I &&idx, double *scratch) {
    double *v = val(th);
    if constexpr (WAVE) {
      double m = v[0];
      m = fmax(m, quad_perm<quad_ctrl(1, 0, 3, 2)>(m));
      m = fmax(m, quad_perm<quad_ctrl(2, 3, 0, 1)>(m));
      m = fmax(m, quad_perm<0x141>(m));
      m = fmax(m, quad_perm<0x140>(m));
      m = fmax(fmax(read_lane(m, 0), read_lane(m, 16)), fmax(read_lane(m, 32), read_lane(m, 48)));
      const unsigned long long who = __ballot(v[0] == m);
      idx(th) = who ? __ffsll((long long)who) - 1 : 0;
      v[0] = m;
    } else {      // every wavefront as above, then the few wavefront results through LDS (a serial scan of blockDim.x entries by every thread cost 5-8 k cycles
      double m = v[0];      //  per call: three of them per pass of the exact mode's active-set method)
      m = fmax(m, quad_perm<quad_ctrl(1, 0, 3, 2)>(m));
      m = fmax(m, quad_perm<quad_ctrl(2, 3, 0, 1)>(m));
      m = fmax(m, quad_perm<0x141>(m));
      m = fmax(m, quad_perm<0x140>(m));
      m = fmax(fmax(read_lane(m, 0), read_lane(m, 16)), fmax(read_lane(m, 32), read_lane(m, 48)));
      const unsigned long long who = __ballot(v[0] == m);
      const int w = threadIdx.x >> 6, nw = (int)(blockDim.x >> 6);
      if ((threadIdx.x & 63) == 0) { scratch[2 * w] = m; scratch[2 * w + 1] = (double)(64 * w + (who ? __ffsll((long long)who) - 1 : 0)); }
      __syncthreads();
      double best = scratch[0];
      int bl = (int)scratch[1];
      for (int k = 1; k < nw; ++k) { const double x = scratch[2 * k]; if (x > best) { best = x; bl = (int)scratch[2 * k + 1]; } }
      __syncthreads();
      v[0] = best; idx(th) = bl;
    }
  }
  template <class V>
  __device__ __forceinline__ void wg_sum(V &&val, double *scratch) {
    double *v = val(th);
    double a = v[0];
    a += quad_perm<quad_ctrl(1, 0, 3, 2)>(a);
    a += quad_perm<quad_ctrl(2, 3, 0, 1)>(a);
    a += quad_perm<0x141>(a);
    a += quad_perm<0x140>(a);
    a = (read_lane(a, 0) + read_lane(a, 16)) + (read_lane(a, 32) + read_lane(a, 48));
    if constexpr (WAVE) v[0] = a;
    else {
      if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = a;
      __syncthreads();
      double tot = scratch[0];
      for (int w = 1; w < (int)(blockDim.x >> 6); ++w) tot += scratch[w];
      __syncthreads();
      v[0] = tot;
    }
  }
  // acc(th)[0 .. 4) += A B for the 16 x 4 matrix A and the 4 x 16 matrix B that the wavefront's lanes hold one element each of (v_mfma_f64_16x16x4_f64:
  // lane l holds A[l & 15][l >> 4] and B[l >> 4][l & 15]; of the 16 x 16 result, element (row (l >> 4) + 4 r, column l & 15) in acc[r]).  The one place of the
  // solver where a matrix instruction pays: the blocked inverse of the seeded Gram matrix (mpc_wrench.h seed_inverse_mfma), one instruction per 16 x 16 tile
  // and four pivots where the scalar sweep spends ~500.
  template <class FA, class FB, class FC>
  __device__ __forceinline__ void mfma16(FA &&fa, FB &&fb, FC &&fc) {
    typedef double v4d __attribute__((ext_vector_type(4)));
    double *c = fc(th);
    v4d acc = {c[0], c[1], c[2], c[3]};
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(fa(th), fb(th), acc, 0, 0, 0);
    c[0] = acc[0]; c[1] = acc[1]; c[2] = acc[2]; c[3] = acc[3];
  }
  // dst(th)[r] <- src(lane r & 3 of the quad)[r >> 2], r = 0 .. 5
  template <class S, class D>
  __device__ __forceinline__ void quad_gather6(S &&src, D &&dst) {
    const double *sv = src(th);
    double *dv = dst(th);
    dv[0] = quad_perm<quad_ctrl(0, 0, 0, 0)>(sv[0]);
    dv[1] = quad_perm<quad_ctrl(1, 1, 1, 1)>(sv[0]);
    dv[2] = quad_perm<quad_ctrl(2, 2, 2, 2)>(sv[0]);
    dv[3] = quad_perm<quad_ctrl(3, 3, 3, 3)>(sv[0]);
    dv[4] = quad_perm<quad_ctrl(0, 0, 0, 0)>(sv[1]);
    dv[5] = quad_perm<quad_ctrl(1, 1, 1, 1)>(sv[1]);
  }
};

}  // namespace mpc
