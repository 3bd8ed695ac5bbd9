// policy_mlp.h -- the weight policy of the reference (RL_Environment/WeightPolicy.py) for N robots:
//   observations (WeightPolicy.compute_observations, :120-139)  ->  actor MLP of rsl_rl's ActorCritic
//   (act_inference = actor(obs): Linear/ELU stack, LeggedCfgPPO.policy: 48 -> 512 -> 256 -> 128 -> 12,
//   RL_Environment/tasks/legged_config_ppo.py:5-9)  ->  clamp to [-1, 1] and the affine map to MPC weights
//   (WeightPolicy.step, :94-118; Parameters.MPC_param_scale / MPC_param_const, MPC_Controller/Parameters.py:25-33).
//
// One fused kernel: a 256-thread workgroup carries 32 robots through every layer; activations stay in LDS,
// weights stream from HBM/L2 (760 KB, shared by all workgroups) as 16-byte loads, the products run on the fp32
// MFMA pipe (v_mfma_f32_32x32x2_f32: exact fp32 FMA chains, so the result differs from a CPU sgemm only by the
// summation order).  Rows of the MFMA tile are robots, columns are output neurons; a wave owns output column
// blocks nb = wave, wave + 4, ...  Inside a chunk of 8 inputs, lane half h = lane / 32 takes inputs 4h .. 4h+3
// over four MFMAs, so both operands are single 16-byte loads.
#pragma once
#include <hip/hip_runtime.h>

namespace policy {

constexpr int kMaxLayers = 8;
constexpr int kRows = 32;            // robots per workgroup (MFMA M)
constexpr int kThreads = 256;
constexpr int kPad = 4;              // floats of row padding in LDS (keeps the 16-byte row reads off one bank)

struct Net {
  int n_layers;
  int dims[kMaxLayers + 1];
  const float *w[kMaxLayers];        // [dims[l+1]][dims[l]] row-major (torch Linear.weight)
  const float *b[kMaxLayers];        // [dims[l+1]]
  float scale[16], shift[16];        // action -> weight map (first dims[n_layers] entries used)
};

typedef float f32x16 __attribute__((ext_vector_type(16)));

// out[r][n] = act( sum_k in[r][k] W[n][k] + b[n] ), r < 32.  K % 8 == 0.
__device__ __forceinline__ void layer(const float *in, int in_stride, float *out, int out_stride, const float *__restrict__ W,
                                      const float *__restrict__ b, int K, int NOUT, bool elu) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, col = lane & 31, h = lane >> 5;
  const int nblocks = (NOUT + 31) / 32;
  if (nblocks >= 2 * (kThreads / 64) && nblocks % 2 == 0) {
    // wide layers: TWO column blocks per wave and trip -- one read of the activations feeds both, and the matrix pipe has two independent accumulation chains,
    // so one block's weight loads travel while the other block's products run (policy step 0.070 -> 0.061 ms per 4096 robots; the products alone
    // are ~24 us).  Every output is still the same chain of fused multiply-adds in the same order: the results do not change by a bit.
    for (int pb = wave; pb < nblocks / 2; pb += kThreads / 64) {
      const int n0 = pb * 64 + col, n1 = n0 + 32;      // (NOUT may end inside the second block)
      const bool live0 = n0 < NOUT, live1 = n1 < NOUT;
      const float *w0 = W + (size_t)(live0 ? n0 : 0) * K + 4 * h, *w1 = W + (size_t)(live1 ? n1 : 0) * K + 4 * h;
      const float *ar = in + col * in_stride + 4 * h;
      f32x16 acc0, acc1;
#pragma unroll
      for (int i = 0; i < 16; ++i) { acc0[i] = 0.f; acc1[i] = 0.f; }
      float4 x0 = *reinterpret_cast<const float4 *>(w0), x1 = *reinterpret_cast<const float4 *>(w1);
      for (int k0 = 0; k0 < K; k0 += 8) {
        const float4 a4 = *reinterpret_cast<const float4 *>(ar + k0);
        float4 u0 = x0, u1 = x1;
        if (k0 + 8 < K) { x0 = *reinterpret_cast<const float4 *>(w0 + k0 + 8); x1 = *reinterpret_cast<const float4 *>(w1 + k0 + 8); }      // the next trip's weights (two trips ahead: no faster)
        if (!live0) u0 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (!live1) u1 = make_float4(0.f, 0.f, 0.f, 0.f);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.x, u0.x, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.x, u1.x, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.y, u0.y, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.y, u1.y, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.z, u0.z, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.z, u1.z, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.w, u0.w, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.w, u1.w, acc1, 0, 0, 0);
      }
#pragma unroll
      for (int blk = 0; blk < 2; ++blk) {
        const int n = blk ? n1 : n0;
        if (blk ? live1 : live0) {
          const float bias = b[n];
#pragma unroll
          for (int reg = 0; reg < 16; ++reg) {
            const int row = (reg & 3) + 8 * (reg >> 2) + 4 * h;
            float v = (blk ? acc1[reg] : acc0[reg]) + bias;
            if (elu) v = v > 0.f ? v : expm1f(v);
            out[row * out_stride + n] = v;
          }
        }
      }
    }
    return;
  }
  for (int nb = wave; nb < nblocks; nb += kThreads / 64) {
    const int n = nb * 32 + col;
    const bool live = n < NOUT;
    const float *wr = W + (size_t)(live ? n : 0) * K + 4 * h;
    const float *ar = in + col * in_stride + 4 * h;     // A operand: row = lane & 31
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    float4 wn = *reinterpret_cast<const float4 *>(wr);
    for (int k0 = 0; k0 < K; k0 += 8) {
      const float4 a4 = *reinterpret_cast<const float4 *>(ar + k0);
      float4 w4 = wn;
      if (k0 + 8 < K) wn = *reinterpret_cast<const float4 *>(wr + k0 + 8);      // the next trip's weights
      if (!live) w4 = make_float4(0.f, 0.f, 0.f, 0.f);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.x, w4.x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.y, w4.y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.z, w4.z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.w, w4.w, acc, 0, 0, 0);
    }
    if (live) {
      const float bias = b[n];
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {   // C layout of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
        const int row = (reg & 3) + 8 * (reg >> 2) + 4 * h;
        float v = acc[reg] + bias;
        if (elu) v = v > 0.f ? v : expm1f(v);            // torch.nn.ELU, alpha = 1
        out[row * out_stride + n] = v;
      }
    }
  }
}

// obs [n, dims[0]] -> actions [n, dims[L]] (may be null) and weights [n, dims[L]] = clamp(a, -1, 1) * scale + shift
__global__ __launch_bounds__(kThreads) void mlp_kernel(Net net, int n, const float *__restrict__ obs, float *__restrict__ actions,
                                                      float *__restrict__ weights) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  int wmax_even = 0, wmax_odd = 0;   // widest activation held by buffer 0 (layers 0, 2, ...) / buffer 1
  for (int l = 0; l <= net.n_layers; ++l) {
    int &m = (l & 1) ? wmax_odd : wmax_even;
    m = net.dims[l] > m ? net.dims[l] : m;
  }
  float *buf[2] = {lds, lds + kRows * (wmax_even + kPad)};
  const int stride[2] = {wmax_even + kPad, wmax_odd + kPad};
  const int r0 = blockIdx.x * kRows, d0 = net.dims[0];
  for (int e = threadIdx.x; e < kRows * d0; e += kThreads) {
    const int r = e / d0, k = e - r * d0;
    buf[0][r * stride[0] + k] = (r0 + r < n) ? obs[(size_t)(r0 + r) * d0 + k] : 0.f;
  }
  __syncthreads();
  for (int l = 0; l < net.n_layers; ++l) {
    layer(buf[l & 1], stride[l & 1], buf[(l + 1) & 1], stride[(l + 1) & 1], net.w[l], net.b[l], net.dims[l], net.dims[l + 1], l + 1 < net.n_layers);
    __syncthreads();
  }
  const int L = net.n_layers, dl = net.dims[L];
  const float *res = buf[L & 1];
  for (int e = threadIdx.x; e < kRows * dl; e += kThreads) {
    const int r = e / dl, k = e - r * dl;
    if (r0 + r >= n) continue;
    const float a = res[r * stride[L & 1] + k];
    if (actions) actions[(size_t)(r0 + r) * dl + k] = a;
    const float c = fminf(fmaxf(a, -1.f), 1.f);          // torch.clamp(current_action, -1, 1); _rescale_actions(-1, 1, .) is the identity
    weights[(size_t)(r0 + r) * dl + k] = c * net.scale[k] + net.shift[k];
  }
}

inline size_t lds_bytes(const Net &net) {
  int we = 0, wo = 0;
  for (int l = 0; l <= net.n_layers; ++l) {
    int &m = (l & 1) ? wo : we;
    m = net.dims[l] > m ? net.dims[l] : m;
  }
  return sizeof(float) * kRows * (size_t)(we + kPad + wo + kPad);
}

// WeightPolicy.compute_observations (:120-139): [vBody * lin, omegaBody * ang, -ground_normal_yaw, commands * (lin, lin, ang),
// dof_pos * dps, dof_vel * dvs, previous actions]  (48 floats).  est = [vBody3, omegaBody3, rpy3, R9] as written by the
// estimator kernel; dof = [12][2] (pos, vel); scales = {lin, ang, dof_pos, dof_vel}.
__global__ void observations_kernel(int n, const float *__restrict__ dof, const float *__restrict__ est, const float *__restrict__ normal,
                                    const float *__restrict__ cmd3, const float *__restrict__ prev, float lin, float ang, float dps, float dvs,
                                    float *__restrict__ obs) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n * 48) return;
  const int r = i / 48, k = i - 48 * r;
  float v;
  if (k < 3) v = est[18 * r + k] * lin;
  else if (k < 6) v = est[18 * r + k] * ang;
  else if (k < 9) v = -normal[3 * r + k - 6];
  else if (k < 12) v = cmd3[3 * r + k - 9] * (k < 11 ? lin : ang);
  else if (k < 24) v = dof[24 * r + 2 * (k - 12)] * dps;
  else if (k < 36) v = dof[24 * r + 2 * (k - 24) + 1] * dvs;
  else v = prev[12 * r + k - 36];
  obs[i] = v;
}

// commands of controller.run: np.concatenate((commands[idx], actions_rescale[idx], [0.0])) (RL_Environment/tasks/aliengo.py:251)
__global__ void pack_commands_kernel(int n, const float *__restrict__ cmd3, const float *__restrict__ w12, float *__restrict__ cmd16) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n * 16) return;
  const int r = i / 16, k = i - 16 * r;
  cmd16[i] = k < 3 ? cmd3[3 * r + k] : k < 15 ? w12[12 * r + k - 3] : 0.f;
}

}  // namespace policy
