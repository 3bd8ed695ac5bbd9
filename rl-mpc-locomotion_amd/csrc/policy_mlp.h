// policy_mlp.h -- the weight policy of the reference (RL_Environment/WeightPolicy.py) for N robots:
//   observations (WeightPolicy.compute_observations, :120-139)  ->  actor MLP of rsl_rl's ActorCritic
//   (act_inference = actor(obs): Linear/ELU stack, LeggedCfgPPO.policy: 48 -> 512 -> 256 -> 128 -> 12,
//   RL_Environment/tasks/legged_config_ppo.py:5-9)  ->  clamp to [-1, 1] and the affine map to MPC weights
//   (WeightPolicy.step, :94-118; Parameters.MPC_param_scale / MPC_param_const, MPC_Controller/Parameters.py:25-33).
//
// One fused kernel: a workgroup carries 16 robots through every layer -- 4096 robots are 256 workgroups, one per CU of the chip (32 robots
// per workgroup left half the CUs idle); activations stay in LDS, weights stream from L2 (760 KB, shared by all workgroups) as 16-byte
// loads, the products run on the fp32 MFMA pipe (v_mfma_f32_16x16x4_f32: exact fp32 FMA chains, so the result differs from a CPU sgemm
// only by the summation order).  Rows of the MFMA tile are robots, columns are output neurons; a wave owns groups of up to four
// 16-column blocks -- four independent accumulation chains (4 VGPRs each) fed by ONE read of the activations, so three blocks' weight
// loads travel while the fourth's products run.  Inside a chunk of 16 inputs, lane quarter q = lane / 16 takes inputs 4q .. 4q+3 over
// four MFMAs, so both operands are single 16-byte loads.
#pragma once
#include <hip/hip_runtime.h>

namespace policy {

constexpr int kMaxLayers = 8;
constexpr int kRows = 16;            // robots per workgroup (MFMA M)
#ifndef POLICY_THREADS
#define POLICY_THREADS 512
#endif
constexpr int kThreads = POLICY_THREADS;
constexpr int kWaves = kThreads / 64;
constexpr int kPad = 4;              // floats of row padding in LDS (keeps the 16-byte row reads off one bank)

struct Net {
  int n_layers;
  int dims[kMaxLayers + 1];
  const float *w[kMaxLayers];        // [dims[l+1]][dims[l]] row-major (torch Linear.weight)
  const float *b[kMaxLayers];        // [dims[l+1]]
  float scale[16], shift[16];        // action -> weight map (first dims[n_layers] entries used)
};

typedef float f32x4 __attribute__((ext_vector_type(4)));

// NB column blocks (16 outputs each) starting at block nb0: out[r][n] = act( sum_k in[r][k] W[n][k] + b[n] ), r < 16.  K % 16 == 0.
template <int NB>
__device__ __forceinline__ void blocks(const float *in, int in_stride, float *out, int out_stride, const float *__restrict__ W,
                                       const float *__restrict__ b, int K, int NOUT, bool elu, int nb0) {
  const int lane = threadIdx.x & 63, col = lane & 15, q = lane >> 4;
  const float *wr[NB];
  bool live[NB];
  f32x4 acc[NB];
  float4 wn[NB];
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    const int n = (nb0 + j) * 16 + col;
    live[j] = n < NOUT;
    wr[j] = W + (size_t)(live[j] ? n : 0) * K + 4 * q;
    acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    wn[j] = *reinterpret_cast<const float4 *>(wr[j]);
  }
  const float *ar = in + col * in_stride + 4 * q;     // A operand: row = lane & 15, k = 4 (lane >> 4) + i in MFMA i of the chunk
  for (int k0 = 0; k0 < K; k0 += 16) {
    const float4 a4 = *reinterpret_cast<const float4 *>(ar + k0);
    float4 w4[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      w4[j] = wn[j];
      if (k0 + 16 < K) wn[j] = *reinterpret_cast<const float4 *>(wr[j] + k0 + 16);      // the next trip's weights
      if (!live[j]) w4[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int j = 0; j < NB; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.x, w4[j].x, acc[j], 0, 0, 0);
#pragma unroll
    for (int j = 0; j < NB; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.y, w4[j].y, acc[j], 0, 0, 0);
#pragma unroll
    for (int j = 0; j < NB; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.z, w4[j].z, acc[j], 0, 0, 0);
#pragma unroll
    for (int j = 0; j < NB; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.w, w4[j].w, acc[j], 0, 0, 0);
  }
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    if (!live[j]) continue;
    const int n = (nb0 + j) * 16 + col;
    const float bias = b[n];
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {   // C layout of the 16x16 MFMA: col = lane & 15, row = 4 (lane >> 4) + reg
      float v = acc[j][reg] + bias;
      if (elu) v = v > 0.f ? v : expm1f(v);            // torch.nn.ELU, alpha = 1
      out[(4 * q + reg) * out_stride + n] = v;
    }
  }
}

__device__ __forceinline__ void layer(const float *in, int in_stride, float *out, int out_stride, const float *__restrict__ W,
                                      const float *__restrict__ b, int K, int NOUT, bool elu) {
  const int wave = threadIdx.x >> 6;
  const int nblocks = (NOUT + 15) / 16;
  // as many blocks per wave and trip as leaves every wave of the workgroup something to do
  if (nblocks >= 4 * kWaves) {
    for (int g = wave; 4 * g < nblocks; g += kWaves) blocks<4>(in, in_stride, out, out_stride, W, b, K, NOUT, elu, 4 * g);
  } else if (nblocks >= 2 * kWaves) {
    for (int g = wave; 2 * g < nblocks; g += kWaves) blocks<2>(in, in_stride, out, out_stride, W, b, K, NOUT, elu, 2 * g);
  } else {
    for (int g = wave; g < nblocks; g += kWaves) blocks<1>(in, in_stride, out, out_stride, W, b, K, NOUT, elu, g);
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
// normal: ground_normal_yaw of robot r at normal[r * normal_stride + 0 .. 2] (3: a packed [n, 3] array; sizeof(CtrlState) / 4: the controller's own state records)
__global__ void observations_kernel(int n, const float *__restrict__ dof, const float *__restrict__ est, const float *__restrict__ normal, int normal_stride,
                                    const float *__restrict__ cmd3, const float *__restrict__ prev, float lin, float ang, float dps, float dvs,
                                    float *__restrict__ obs) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n * 48) return;
  const int r = i / 48, k = i - 48 * r;
  float v;
  if (k < 3) v = est[18 * r + k] * lin;
  else if (k < 6) v = est[18 * r + k] * ang;
  else if (k < 9) v = -normal[(size_t)normal_stride * r + k - 6];
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

// ... with the rescale of VecTask.pre_physics_step in front of it: actions_rescale = torch.mul(actions, MPC_param_scale).add(MPC_param_const)
// (RL_Environment/tasks/aliengo.py:237-245) -- a float32 product, then a float32 sum, like torch's two kernels (no fused multiply-add)
struct Rescale { float scale[12], shift[12]; };
__global__ void pack_commands_scaled_kernel(int n, const float *__restrict__ cmd3, const float *__restrict__ act12, Rescale rs, float *__restrict__ cmd16) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n * 16) return;
  const int r = i / 16, k = i - 16 * r;
  float v = 0.f;
  if (k < 3) v = cmd3[3 * r + k];
  else if (k < 15) v = __fadd_rn(__fmul_rn(act12[12 * r + k - 3], rs.scale[k - 3]), rs.shift[k - 3]);
  cmd16[i] = v;
}

}  // namespace policy
