// mpc_batch.hip -- gfx950 kernels + the C ABI of include/mpc_batch.h.
// One workgroup per robot and kernel: assembly and Ruiz scaling (mpc_core.h, dense P in register tiles), then the OSQP
// iteration in the wrench space (mpc_wrench.h; one wavefront per robot at h = 10).  All arithmetic fp64.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#define MPC_LOCKSTEP 1   // a single-wavefront workgroup executes its LDS instructions in program order (mpc_wrench.h: Shared::NBUF)
#include "../../include/mpc_batch.h"
#include "controller.h"
#include "mpc_core.h"
#include "mpc_model.h"
#include "mpc_wrench.h"
#include "policy_mlp.h"

using namespace mpc;

// minimum waves per SIMD the register allocator must leave room for in the prep kernel: 3 -> a 256-thread workgroup (h = 10)
// gets 168 VGPRs and three robots share a CU.  (Four -- 128 VGPRs, 40 KB of LDS each -- were measured slower: 0.278 against
// 0.235 ms per 4096 robots; the Ruiz passes then spill.)
#ifndef MPC_SCALE_MIN_WAVES
#define MPC_SCALE_MIN_WAVES 3
#endif
#ifndef MPC_MIN_WAVES_MAX_T
#define MPC_MIN_WAVES_MAX_T 256   // larger workgroups (h = 16), and the four-tiles-per-thread layout (h = 20), run one per CU
#endif
// waves per SIMD of the solve kernel (h = 10: one wave per robot).  1 -> the full 512-register budget (AGPRs as spill space), four
// robots per CU: measured faster than two waves per SIMD at 256 registers, which spills to scratch memory
#ifndef MPC_SOLVE_MIN_WAVES
#define MPC_SOLVE_MIN_WAVES 1
#endif
#ifndef MPC_SOLVE_MIN_WAVES_WIDE   // the multi-wave workgroups of the long horizons (128 threads at h = 12, 256 at h = 16 / 20): two waves per
#define MPC_SOLVE_MIN_WAVES_WIDE 2  // SIMD hide their barriers (measured: h = 16 3.00 -> 2.35 ms, h = 20 3.55 -> 2.83 ms per 4096 robots)
#endif

namespace {

thread_local std::string g_err;
int fail(int code, const std::string &msg) { g_err = msg; return code; }
// Every entry point works on the device its handle was created on and leaves the caller's current device as it found it
// (a process may hold handles on several GPUs, and the caller -- torch -- has a current device of its own).
struct DeviceGuard {
  int prev = -1;
  bool switched = false;
  explicit DeviceGuard(int dev) {
    if (hipGetDevice(&prev) == hipSuccess && prev != dev) switched = hipSetDevice(dev) == hipSuccess;
  }
  ~DeviceGuard() { if (switched) (void)hipSetDevice(prev); }
  DeviceGuard(const DeviceGuard &) = delete;
  DeviceGuard &operator=(const DeviceGuard &) = delete;
};
#define HIP_TRY(expr)                                                                          \
  do {                                                                                         \
    hipError_t e_ = (expr);                                                                    \
    if (e_ != hipSuccess) return fail(MPC_E_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
  } while (0)

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
    } else {
      scratch[threadIdx.x] = v[0];
      __syncthreads();
      double m = scratch[0];
      int ml = 0;
      for (int i = 1; i < (int)blockDim.x; ++i) { const double x = scratch[i]; if (x > m) { m = x; ml = i; } }
      __syncthreads();
      v[0] = m; idx(th) = ml;
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

// The planning horizons compiled into the library (ConvexMpc accepts any planning_horizon, mpc_osqp.cc:186-190, 508-574; the shipped
// Python uses 10, ConvexMPCLocomotion.py:27; BASELINE's configurations 10, 16, 20).  Every entry instantiates the prep, solve (job),
// exact and fall-back kernels for that horizon: -DMPC_HORIZON_LIST to build another set (any h >= 2 whose workgroups fit: h <= 20).
#ifndef MPC_HORIZON_LIST
#define MPC_HORIZON_LIST(X) X(8) X(10) X(12) X(16) X(20)
#endif

constexpr int kSchedNext = 0, kSchedHead = 1, kSchedTail = 2, kSchedJobs = 3, kSchedLen = 4;   // job bookkeeping of a launch (mpc_solve_jobs_kernel)

// Solve kernel (mpc_wrench.h Solver): ADMM + polish of every active robot, from the QP and scale records.  EXACT: the
// exact-optimum mode (the reference's qpOASES branch) -- a separate instantiation, so that its outer loop does not touch the
// register allocation of the OSQP mode.
template <int H, bool EXACT>
__global__ __launch_bounds__(Cfg<H>::TW, (Cfg<H>::TW <= 64 ? MPC_SOLVE_MIN_WAVES : MPC_SOLVE_MIN_WAVES_WIDE)) void mpc_solve_kernel(
    int n, const RobotModel *__restrict__ models, double *__restrict__ state, const double *__restrict__ qp,
    const double *__restrict__ sc, double *__restrict__ forces, int *__restrict__ info, long long *__restrict__ prof,
    const int *__restrict__ order, const int *__restrict__ sched, const int *__restrict__ ready, int max_iter) {
  // static LDS: absolute addresses fold into the ds_* offset fields
  __shared__ __attribute__((aligned(16))) Shared<H> sh;
  using C = Cfg<H>;
  // OSQP mode: the job list holds the launch's active robots (robots whose controller is between two MPC updates have no job), longest
  // expected solve first (order_block).  Exact mode: this kernel is the second launch -- the robots whose active set mpc_exact_kernel
  // could not certify, listed in `ready` -- and takes the ADMM route.
  if ((int)blockIdx.x >= (EXACT ? sched[kSchedTail] : sched[kSchedJobs])) return;
  const int robot = EXACT ? ready[blockIdx.x] : order[blockIdx.x];
  WThread<H> th;
  th.init(threadIdx.x);
#pragma unroll
  for (int j = 0; j < C::TE; ++j) th.Mx[j] = 0;
  using Ex = DeviceExec<WThread<H>, (C::TW <= 64)>;
  Ex ex{th};
  const RobotModel &mdl = models[robot];   // (uniform loads; a by-value copy indexed at run time would sit in scratch)
  Solver<H, Ex> sv{ex,
                                       sh,
                                       mdl,
                                       state + (size_t)robot * state_len<H>(),
                                       qp + (size_t)robot * C::QP_LEN,
                                       sc + (size_t)robot * C::SC_LEN,
                                       forces + (size_t)robot * C::N,
                                       info + (size_t)robot * kInfoLen,
                                       prof ? prof + (size_t)robot * kProfLen : nullptr};
  if constexpr (EXACT) sv.exact();
  else sv.max_iter = max_iter;
  sv.template run<EXACT>();
}

// Exact mode (the reference's qpOASES branch), first launch: the dual active-set method + the polish on its set (mpc_wrench.h
// active_set / run_active_set), one workgroup per robot.  A robot whose set is not certified (the polished point fails the optimality
// test, the working set overflows its slots) is appended to `ready` for the second launch, mpc_solve_kernel<H, true>.
template <int H>
__global__ __launch_bounds__(Cfg<H>::TW, (Cfg<H>::TW <= 64 ? MPC_SOLVE_MIN_WAVES : MPC_SOLVE_MIN_WAVES_WIDE)) void mpc_exact_kernel(
    const RobotModel *__restrict__ models, double *__restrict__ state, const double *__restrict__ qp, const double *__restrict__ sc,
    double *__restrict__ forces, int *__restrict__ info, long long *__restrict__ prof, const int *__restrict__ order, int *__restrict__ sched,
    int *__restrict__ ready) {
  __shared__ __attribute__((aligned(16))) Shared<H> sh;
  __shared__ __attribute__((aligned(16))) GiShared<H> gsh;
  using C = Cfg<H>;
  if ((int)blockIdx.x >= sched[kSchedJobs]) return;
  const int robot = order[blockIdx.x];
  WThread<H> th;
  th.init(threadIdx.x);
#pragma unroll
  for (int j = 0; j < C::TE; ++j) th.Mx[j] = 0;
  using Ex = DeviceExec<WThread<H>, (C::TW <= 64)>;
  Ex ex{th};
  Solver<H, Ex> sv{ex, sh, models[robot], state + (size_t)robot * state_len<H>(), qp + (size_t)robot * C::QP_LEN, sc + (size_t)robot * C::SC_LEN,
                   forces + (size_t)robot * C::N, info + (size_t)robot * kInfoLen, prof ? prof + (size_t)robot * kProfLen : nullptr};
  sv.exact();
  sv.gi = &gsh;
  const bool ok = sv.run_active_set();
  if (!ok && threadIdx.x == 0) ready[atomicAdd(&sched[kSchedTail], 1)] = robot;
}

// The OSQP-mode solve as a PERSISTENT kernel: one workgroup per wave slot of the chip, each pulling jobs until none is left.  A solve
// is two jobs (mpc_wrench.h admm_job / polish_job): the ADMM part, 25 to 250+ iterations long, and the polish, the same ~100 k cycles for
// every robot and dependent on the ADMM part's result only (x, z, y in the state record, two residuals).  With one job per robot a
// 4096-robot launch is four jobs of very different length per wave slot, and the launch ends when the unluckiest slot does
// (measured 0.69 ms against 0.55 ms of work per slot, tools/sched_model.py); with the polishes as uniform filler jobs -- taken only
// once no ADMM job is left to start -- the tail shrinks to a fraction of one polish.
//   sched[kSchedNext]  next ADMM job (index into `order`)          sched[kSchedTail]  polish entries published
//   sched[kSchedHead]  next polish entry to take                   sched[kSchedJobs]  number of jobs (active robots; order_block)
//   ready[i]           -1 not yet published; robot: polish it; -2: that solve needs no polish (not SOLVED)
// An ADMM job publishes exactly one entry, in completion order, after a device-scope release of its results; a wave that takes entry
// i spins until it is there (every job of the launch is then running or done, so the wait is bounded by the longest ADMM part)
// and acquires before it loads the record.
template <int H>
__global__ __launch_bounds__(Cfg<H>::TW, (Cfg<H>::TW <= 64 ? MPC_SOLVE_MIN_WAVES : MPC_SOLVE_MIN_WAVES_WIDE)) void mpc_solve_jobs_kernel(
    const RobotModel *__restrict__ models, double *__restrict__ state, const double *__restrict__ qp, double *__restrict__ sc,
    double *__restrict__ forces, int *__restrict__ info, long long *__restrict__ prof, const int *__restrict__ order, int *__restrict__ sched,
    int *__restrict__ ready, int max_iter) {
  __shared__ __attribute__((aligned(16))) Shared<H> sh;
  __shared__ int job;
  using C = Cfg<H>;
  using Ex = DeviceExec<WThread<H>, (C::TW <= 64)>;
  WThread<H> th;
  th.init(threadIdx.x);
  Ex ex{th};
  const int njobs = sched[kSchedJobs];
  auto solver = [&](int robot) {
    return Solver<H, Ex>{ex, sh, models[robot], state + (size_t)robot * state_len<H>(), qp + (size_t)robot * C::QP_LEN, sc + (size_t)robot * C::SC_LEN,
                         forces + (size_t)robot * C::N, info + (size_t)robot * kInfoLen, prof ? prof + (size_t)robot * kProfLen : nullptr};
  };
  for (;;) {     // ---- ADMM jobs, in the dispatch order of order_block
    [[maybe_unused]] const long long tf0 = MPC_CLOCK();
    ex.par([&](WThread<H> &t) { if (t.tid == 0) job = atomicAdd(&sched[kSchedNext], 1); });
    const int idx = job;
    ex.par([](WThread<H> &) {});     // (everybody has read `job` before thread 0 overwrites it)
    if (idx >= njobs) break;
    const int robot = order[idx];
#pragma unroll
    for (int j = 0; j < C::TE; ++j) th.Mx[j] = 0;
    bool pol;
    {
      Solver<H, Ex> sv = solver(robot);
      sv.max_iter = max_iter;
      sv.jobrec = sc + (size_t)robot * C::SC_LEN + C::SC_JOB;
      pol = sv.admm_job();
      if (MPC_PROFILE_SUB == 8 && prof && threadIdx.x == 0) MPC_GST(prof + (size_t)robot * kProfLen + 1, (long long)(sv.t_start - tf0));
    }
    // the job's results are device-coherent stores (MPC_GST): once they have completed -- a workgroup-scope release is the wait for
    // that, with no L2 write-back -- the entry may be published
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    ex.par([&](WThread<H> &t) {
      if (t.tid == 0) {
        const int pos = atomicAdd(&sched[kSchedTail], 1);
        __hip_atomic_store(&ready[pos], pol ? robot : -2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    });
  }
  for (;;) {     // ---- polish jobs, in completion order of the ADMM parts
    [[maybe_unused]] const long long tf0 = MPC_CLOCK();
    ex.par([&](WThread<H> &t) {
      if (t.tid == 0) {
        const int pos = atomicAdd(&sched[kSchedHead], 1);
        int e = -2;
        if (pos < njobs) {
          while ((e = __hip_atomic_load(&ready[pos], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == -1) __builtin_amdgcn_s_sleep(32);
        } else e = -3;
        job = e;
      }
    });
    const int e = job;
    ex.par([](WThread<H> &) {});
    if (e == -3) break;
    if (e < 0) continue;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");   // (the ADMM job's results are read with device-coherent loads, MPC_GLD: no L2 invalidate)
#pragma unroll
    for (int j = 0; j < C::TE; ++j) th.Mx[j] = 0;
    Solver<H, Ex> sv = solver(e);
    sv.jobrec = sc + (size_t)e * C::SC_LEN + C::SC_JOB;
    sv.polish_job();
    if (MPC_PROFILE_SUB == 8 && prof && threadIdx.x == 0) MPC_GST(prof + (size_t)e * kProfLen + 2, (long long)(sv.t_start - tf0));
  }
}

// Workgroup -> robot order for the solve kernel of THIS launch: robots sorted by the shader cycles their previous solve took,
// longest first.  Runs as one extra workgroup of the assembly kernel (blockIdx.x == 0), i.e. hidden behind the assembly.
// Warm-started robots repeat their iteration counts from step to step, and solve times differ 3x between a 25-iteration
// and a 75-iteration robot; dispatching the long ones first keeps the tail of the launch short (a counting sort over
// cycles / 16384 in one workgroup; the order inside a bucket is arbitrary, results do not depend on it).
// The sort key is the LONGEST of the robot's last kOrderHistory solves (one byte each, cycles / 16384): the reference's gaits
// have ten segments, so a robot's hard phases (touch-down, lift-off) recur every ten solves, and a solve that is queued as
// short but runs long is what stretches the tail (tools/tail_model.py: ordering by the previous solve alone 0.717 ms per
// launch on average, by this key 0.692, clairvoyant 0.650).
constexpr int kOrderBuckets = 256, kOrderHistory = 10;
// Also the launch's job bookkeeping: only ACTIVE robots (active == null: all) enter the list, sched[kSchedJobs] = their number, the
// job counters and the polish entries of mpc_solve_jobs_kernel are reset.
__device__ void order_block(int n, const long long *__restrict__ prof, unsigned char *__restrict__ hist, int slot, int *__restrict__ order,
                            const int *__restrict__ active, int *__restrict__ sched, int *__restrict__ ready) {
  __shared__ int cnt[kOrderBuckets], base[kOrderBuckets];
  __shared__ int filled;
  for (int b = threadIdx.x; b < kOrderBuckets; b += blockDim.x) cnt[b] = 0;
  for (int r = threadIdx.x; r < n; r += blockDim.x) ready[r] = -1;
  if (threadIdx.x == 0) filled = 0;
  __syncthreads();
  constexpr int kPer = 8;                         // robots per thread held in registers (n <= 8192 per pass)
  for (int r0 = 0; r0 < n; r0 += kPer * blockDim.x) {
    int bk[kPer], rank[kPer];
#pragma unroll
    for (int i = 0; i < kPer; ++i) {
      const int r = r0 + i * blockDim.x + threadIdx.x;
      bk[i] = -1;
      if (r < n && (!active || active[r])) {
        const long long c = prof[(size_t)r * kProfLen + kProfLen - 1] >> 14;
        unsigned char *hr = hist + (size_t)r * kOrderHistory;
        hr[slot] = (unsigned char)(c < 0 ? 0 : (c >= kOrderBuckets ? kOrderBuckets - 1 : c));
        int mx = 0;
#pragma unroll
        for (int k = 0; k < kOrderHistory; ++k) mx = max(mx, (int)hr[k]);
        bk[i] = mx;
        rank[i] = atomicAdd(&cnt[bk[i]], 1);      // rank inside the bucket (within this pass)
      }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      int acc = filled;                           // earlier passes fill the front of `order` (only n > 8192 has several)
      for (int b = kOrderBuckets - 1; b >= 0; --b) { base[b] = acc; acc += cnt[b]; cnt[b] = 0; }
      filled = acc;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < kPer; ++i)
      if (bk[i] >= 0) order[base[bk[i]] + rank[i]] = r0 + i * blockDim.x + threadIdx.x;
    __syncthreads();
  }
  if (threadIdx.x == 0) { sched[kSchedNext] = 0; sched[kSchedHead] = 0; sched[kSchedTail] = 0; sched[kSchedJobs] = filled; }
}

// Prep kernel (mpc_core.h Assembler + Scaler): QP record (q, bounds, cone block, wrench form of P) and scale record (OSQP's Ruiz
// equilibration) of every active robot.  The dense P lives only in this kernel's registers.
template <int H>
__global__ __launch_bounds__(Cfg<H>::T, (Cfg<H>::T <= MPC_MIN_WAVES_MAX_T && Cfg<H>::NT == 1 ? MPC_SCALE_MIN_WAVES : 1)) void mpc_prep_kernel(
    int n, const RobotModel *__restrict__ models, const float *__restrict__ in, const double *__restrict__ in64, const double *__restrict__ state,
    double *__restrict__ qp, double *__restrict__ sc, long long *__restrict__ prof, const int *__restrict__ active, int *__restrict__ order,
    unsigned char *__restrict__ hist, int hist_slot, int *__restrict__ sched, int *__restrict__ ready) {
  __shared__ __attribute__((aligned(16))) PrepShared<H> sh;
  using C = Cfg<H>;
  if (blockIdx.x == 0) {      // the extra workgroup (first, so that it starts at once): job list / dispatch order of the solve kernel that follows
    order_block(n, prof, hist, hist_slot, order, active, sched, ready);
    return;
  }
  const int robot = (int)blockIdx.x - 1;
  if (robot >= n) return;
  if (active && !active[robot]) return;
  Thread<H> th;
  th.init(threadIdx.x);
#pragma unroll
  for (int j = 0; j < C::NT * C::TE; ++j) th.Mx[j] = 0;
  using Ex = DeviceExec<Thread<H>>;
  Ex ex{th};
  const RobotModel &mdl = models[robot];
  double *qpr = qp + (size_t)robot * C::QP_LEN;
  Assembler<H, Ex> am{ex, sh.as, mdl, in ? in + (size_t)robot * C::IN_LEN : nullptr, in64 ? in64 + (size_t)robot * C::IN_LEN : nullptr, sh.u12, qpr, prof ? prof + (size_t)robot * kProfLen : nullptr};
  am.run();
  Scaler<H, Ex> sk{ex, sh.sc, state + (size_t)robot * state_len<H>(), sh.u12, mdl.alpha, qpr, sc + (size_t)robot * C::SC_LEN, prof ? prof + (size_t)robot * kProfLen : nullptr};
  sk.run();
}

__global__ void reset_kernel(double *state, int state_len, const int *ids, int k, int n) {
  const int r = blockIdx.x;
  const int robot = ids ? ids[r] : r;
  if (r >= k || robot < 0 || robot >= n) return;
  for (int i = threadIdx.x; i < state_len; i += blockDim.x) state[(size_t)robot * state_len + i] = 0.0;
}

template <int H>
int launch(int n, const RobotModel *models, const float *in, const double *in64, double *state, double *qp, double *sc, double *forces, int *info,
           long long *prof, const int *active, int *order, unsigned char *hist, int hist_slot, hipEvent_t *ev, hipStream_t stream, int exact, int max_iter,
           int *sched, int *ready, int job_slots) {
  if (ev) (void)hipEventRecord(ev[0], stream);
  hipLaunchKernelGGL(mpc_prep_kernel<H>, dim3(n + 1), dim3(Cfg<H>::T), 0, stream, n, models, in, in64, state, qp, sc, prof, active, order, hist, hist_slot, sched, ready);
  if (ev) (void)hipEventRecord(ev[1], stream);
  if (exact) {
    hipLaunchKernelGGL((mpc_exact_kernel<H>), dim3(n), dim3(Cfg<H>::TW), 0, stream, models, state, qp, sc, forces, info, prof, order, sched, ready);
    hipLaunchKernelGGL((mpc_solve_kernel<H, true>), dim3(n), dim3(Cfg<H>::TW), 0, stream, n, models, state, qp, sc, forces, info, prof, order, sched, ready, max_iter);
  }
  else if (job_slots > 0) {   // persistent workgroups, ADMM and polish as separate jobs (h = 16: 2.68 -> 2.28 ms, h = 20: 3.36 -> 2.78 ms per 4096 robots)
    const int slots = Cfg<H>::TW <= 64 ? job_slots : job_slots / 2;         // (multi-wave workgroups: two per CU)
    hipLaunchKernelGGL((mpc_solve_jobs_kernel<H>), dim3(n < slots ? n : slots), dim3(Cfg<H>::TW), 0, stream, models, state, qp, sc, forces, info, prof, order, sched, ready, max_iter);
  }
  else hipLaunchKernelGGL((mpc_solve_kernel<H, false>), dim3(n), dim3(Cfg<H>::TW), 0, stream, n, models, state, qp, sc, forces, info, prof, order, sched, ready, max_iter);
  if (ev) (void)hipEventRecord(ev[2], stream);
  HIP_TRY(hipGetLastError());
  return MPC_OK;
}


}  // namespace

constexpr int kTimingRing = 64;

struct mpc_batch {
  int n = 0, h = 0;
  int state_len = 0;
  RobotModel *d_models = nullptr;
  double *d_state = nullptr, *d_qp = nullptr, *d_sc = nullptr;   // warm start, QP record (q, l, u, cone, wrench form of P), scale record
  int *d_info = nullptr;   // used when the caller passes no info buffer
  long long *d_prof = nullptr;   // per-robot section cycle counts of the last solve
  int *d_order = nullptr;        // job list of the solve kernel: the launch's active robots, longest expected solve first (order_block, written by the prep launch)
  int *d_sched = nullptr;        // [kSchedLen] job counters of the launch; d_ready [n]: polish entries (mpc_solve_jobs_kernel)
  int *d_ready = nullptr;
  int job_slots = 0;             // wave slots of the device for the persistent job kernel (0: one workgroup per robot)
  bool timing = false;           // mpc_batch_enable_timing: HIP events around the two kernels of each launch
  hipEvent_t ev[kTimingRing][3];
  long long launches = 0;
  float *d_host_in = nullptr;    // staging for mpc_batch_solve_host
  double *d_host_in64 = nullptr; // ... and mpc_batch_solve_host_f64
  double *d_host_f = nullptr;
  unsigned char *d_hist = nullptr;   // [n][kOrderHistory] cycles / 16384 of the last solves (order_block's sort key is their maximum)
  unsigned long long order_launches = 0;
  int device = 0;                // the HIP device the handle was created on: every entry point runs under a DeviceGuard for it
  int exact = 0;                 // mpc_batch_set_solver: 1 = the QP's exact optimum (the reference's qpOASES branch), cold on every call
  int max_iter = kMaxIter;       // mpc_batch_set_max_iter: OSQP's max_iter setting (OSQP mode)
  long long bytes = 0;
};


// one solver launch on b's robots (+ the dispatch order for the next one)
static int launch_solver(mpc_batch *b, const float *d_in, double *d_forces, int *d_info, const int *d_active, hipStream_t st, const double *d_in64 = nullptr) {
  DeviceGuard guard_(b->device);
  const int slot = (int)(b->order_launches++ % kOrderHistory);
  if (b->exact) HIP_TRY(hipMemsetAsync(b->d_state, 0, sizeof(double) * (size_t)b->n * b->state_len, st));   // no warm start in that branch (mpc_osqp.cc:906-919)
  hipEvent_t *ev = b->timing ? b->ev[b->launches % kTimingRing] : nullptr;
  int rc = MPC_E_HORIZON;
#define MPC_LAUNCH(HH) launch<HH>(b->n, b->d_models, d_in, d_in64, b->d_state, b->d_qp, b->d_sc, d_forces, d_info, b->d_prof, d_active, b->d_order, b->d_hist, slot, ev, st, b->exact, b->max_iter, b->d_sched, b->d_ready, b->job_slots)
  switch (b->h) {
#define MPC_CASE(HH) case HH: rc = MPC_LAUNCH(HH); break;
    MPC_HORIZON_LIST(MPC_CASE)
#undef MPC_CASE
  }
#undef MPC_LAUNCH
  if (rc == MPC_E_HORIZON) return fail(MPC_E_HORIZON, "solver launch: horizon not compiled in");
  if (rc != MPC_OK) return rc;
  b->launches++;
  return MPC_OK;
}

#define MPC_QP(HH) if (h == HH) return Cfg<HH>::QP_LEN;
#define MPC_SC(HH) if (h == HH) return Cfg<HH>::SC_LEN;
#define MPC_XQP(HH) if (h == HH) return Cfg<HH>::XQP_LEN;
#define MPC_XSC(HH) if (h == HH) return Cfg<HH>::XSC_LEN;
static size_t qp_len_of(int h) { MPC_HORIZON_LIST(MPC_QP) return 0; }     // (0: horizon not compiled in)
static size_t sc_len_of(int h) { MPC_HORIZON_LIST(MPC_SC) return 0; }
static size_t xqp_len_of(int h) { MPC_HORIZON_LIST(MPC_XQP) return 0; }   // the records as the accessors hand them out
static size_t xsc_len_of(int h) { MPC_HORIZON_LIST(MPC_XSC) return 0; }
#undef MPC_QP
#undef MPC_SC
#undef MPC_XQP
#undef MPC_XSC
// The device records keep three bound values and nine cone entries per foot (mpc_core.h); the accessors hand out every bound and the
// dense cone block, formed here exactly as the solve kernel forms them (E times the bound).
template <int H>
static void expand_records(const double *qp, const double *sc, double *xqp, double *xsc) {
  using C = Cfg<H>;
  if (xqp) {
    for (int i = 0; i < C::N; ++i) xqp[i] = qp[C::QP_Q + i];
    for (int f = 0; f < C::NF; ++f)
      for (int r = 0; r < 5; ++r) {
        xqp[C::XQP_L + 5 * f + r] = r < 4 ? 0.0 : qp[C::QP_BND + 3 * f];
        xqp[C::XQP_U + 5 * f + r] = r < 4 ? qp[C::QP_BND + 3 * f + 1] : qp[C::QP_BND + 3 * f + 2];
      }
    for (int i = 0; i < 16 + 72 + 36 + 8; ++i) xqp[C::XQP_CONE + i] = qp[C::QP_CONE + i];
  }
  if (xsc) {
    for (int i = 0; i < 2 * C::N + C::M; ++i) xsc[i] = sc[i];      // D, E, q_s
    for (int f = 0; f < C::NF; ++f) {
      for (int k = 0; k < 15; ++k) xsc[C::XSC_AS + 15 * f + k] = 0.0;
      for (int k = 0; k < 9; ++k) xsc[C::XSC_AS + 15 * f + kAsPos[k]] = sc[C::SC_AS + 9 * f + k];
      for (int r = 0; r < 5; ++r) {
        const double e = sc[C::SC_E + 5 * f + r];
        xsc[C::XSC_LS + 5 * f + r] = e * (r < 4 ? 0.0 : qp[C::QP_BND + 3 * f]);
        xsc[C::XSC_US + 5 * f + r] = e * (r < 4 ? qp[C::QP_BND + 3 * f + 1] : qp[C::QP_BND + 3 * f + 2]);
      }
    }
    for (int i = 0; i < 4; ++i) xsc[C::XSC_C + i] = sc[C::SC_C + i];
  }
}
static int fetch_records(mpc_batch *b, double *h_qp, double *h_sc) {
  const size_t ql = qp_len_of(b->h), sl = sc_len_of(b->h), xql = xqp_len_of(b->h), xsl = xsc_len_of(b->h);
  std::vector<double> qp((size_t)b->n * ql), sc((size_t)b->n * sl);
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpy(qp.data(), b->d_qp, sizeof(double) * qp.size(), hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(sc.data(), b->d_sc, sizeof(double) * sc.size(), hipMemcpyDeviceToHost));
  for (int r = 0; r < b->n; ++r) {
    const double *q = qp.data() + (size_t)r * ql, *c = sc.data() + (size_t)r * sl;
    double *xq = h_qp ? h_qp + (size_t)r * xql : nullptr, *xs = h_sc ? h_sc + (size_t)r * xsl : nullptr;
#define MPC_X(HH) if (b->h == HH) expand_records<HH>(q, c, xq, xs);
    MPC_HORIZON_LIST(MPC_X)
#undef MPC_X
  }
  return MPC_OK;
}

extern "C" {

const char *mpc_last_error(void) { return g_err.c_str(); }
int mpc_input_len(int horizon) { return 56 + 4 * horizon; }
int mpc_supported_horizons(int *out, int cap) {
#define MPC_ITEM(HH) HH,
  const int hs[] = {MPC_HORIZON_LIST(MPC_ITEM)};
#undef MPC_ITEM
  const int cnt = (int)(sizeof hs / sizeof *hs);
  for (int i = 0; i < cnt && i < cap; ++i) out[i] = hs[i];
  return cnt;
}

int mpc_batch_create(mpc_batch **out, int n, int horizon, double timestep, double alpha, const double *mass,
                     const double *inertia9) {
  if (!out || n <= 0 || !mass || !inertia9) return fail(MPC_E_ARG, "mpc_batch_create: bad argument");
  if (qp_len_of(horizon) == 0) return fail(MPC_E_HORIZON, "mpc_batch_create: planning horizon not compiled in (see mpc_supported_horizons; -DMPC_HORIZON_LIST builds another set)");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail(MPC_E_NODEVICE, "mpc_batch_create: no HIP device");
  mpc_batch *b = new mpc_batch();
  (void)hipGetDevice(&b->device);
  b->n = n;
  b->h = horizon;
  const size_t qp_len = qp_len_of(horizon), sc_len = sc_len_of(horizon);
  b->state_len = (int)(64 * horizon + 2);
  std::vector<RobotModel> models(n);
  for (int i = 0; i < n; ++i) models[i] = make_model(mass[i], inertia9 + 9 * (size_t)i, timestep, alpha);
  auto cleanup = [&]() { mpc_batch_destroy(b); };
  hipError_t e;
  if ((e = hipMalloc(&b->d_models, sizeof(RobotModel) * n)) != hipSuccess ||
      (e = hipMalloc(&b->d_state, sizeof(double) * (size_t)n * b->state_len)) != hipSuccess ||
      (e = hipMalloc(&b->d_qp, sizeof(double) * (size_t)n * qp_len)) != hipSuccess ||
      (e = hipMalloc(&b->d_sc, sizeof(double) * (size_t)n * sc_len)) != hipSuccess ||
      (e = hipMalloc(&b->d_info, sizeof(int) * (size_t)n * kInfoLen)) != hipSuccess ||
      (e = hipMalloc(&b->d_prof, sizeof(long long) * (size_t)n * kProfLen)) != hipSuccess ||
      (e = hipMalloc(&b->d_order, sizeof(int) * (size_t)n)) != hipSuccess ||
      (e = hipMalloc(&b->d_hist, (size_t)n * kOrderHistory)) != hipSuccess ||
      (e = hipMalloc(&b->d_sched, sizeof(int) * kSchedLen)) != hipSuccess ||
      (e = hipMalloc(&b->d_ready, sizeof(int) * (size_t)n)) != hipSuccess ||
      (e = hipMemset(b->d_prof, 0, sizeof(long long) * (size_t)n * kProfLen)) != hipSuccess ||
      (e = hipMemcpy(b->d_models, models.data(), sizeof(RobotModel) * n, hipMemcpyHostToDevice)) != hipSuccess ||
      (e = hipMemset(b->d_hist, 0, (size_t)n * kOrderHistory)) != hipSuccess ||
      (e = hipMemset(b->d_state, 0, sizeof(double) * (size_t)n * b->state_len)) != hipSuccess) {
    cleanup();
    return fail(MPC_E_HIP, std::string("mpc_batch_create: ") + hipGetErrorString(e));
  }
  {   // the persistent job kernel runs one workgroup per wave slot: four single-wave workgroups per CU (one per SIMD, full register budget)
    hipDeviceProp_t prop;
    const char *env = getenv("MPC_SOLVE_JOBS");      // tuning hook: 0 = one workgroup per robot (the round-2 launch), N > 0 = that many slots
    if (hipGetDeviceProperties(&prop, b->device) == hipSuccess) b->job_slots = 4 * prop.multiProcessorCount;
    if (env) b->job_slots = atoi(env);
  }
  b->bytes = (long long)(sizeof(RobotModel) * n + sizeof(double) * (size_t)n * (b->state_len + qp_len + sc_len) + sizeof(int) * (size_t)n * kInfoLen);
  *out = b;
  return MPC_OK;
}

void mpc_batch_destroy(mpc_batch *b) {
  if (!b) return;
  if (b->d_models) (void)hipFree(b->d_models);
  if (b->d_state) (void)hipFree(b->d_state);
  if (b->d_qp) (void)hipFree(b->d_qp);
  if (b->d_sc) (void)hipFree(b->d_sc);
  if (b->d_info) (void)hipFree(b->d_info);
  if (b->d_prof) (void)hipFree(b->d_prof);
  if (b->d_order) (void)hipFree(b->d_order);
  if (b->d_hist) (void)hipFree(b->d_hist);
  if (b->d_sched) (void)hipFree(b->d_sched);
  if (b->d_ready) (void)hipFree(b->d_ready);
  if (b->timing) for (auto &e3 : b->ev) for (auto &e : e3) (void)hipEventDestroy(e);
  if (b->d_host_in) (void)hipFree(b->d_host_in);
  if (b->d_host_in64) (void)hipFree(b->d_host_in64);
  if (b->d_host_f) (void)hipFree(b->d_host_f);
  delete b;
}

int mpc_batch_solve(mpc_batch *b, const float *d_in, double *d_forces, int *d_info, void *stream) {
  if (!b || !d_in || !d_forces) return fail(MPC_E_ARG, "mpc_batch_solve: bad argument");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  int *info = d_info ? d_info : b->d_info;
  return launch_solver(b, d_in, d_forces, info, nullptr, st);
}

int mpc_batch_set_solver(mpc_batch *b, int solver) {
  if (!b || (solver != MPC_SOLVER_OSQP && solver != MPC_SOLVER_EXACT)) return fail(MPC_E_ARG, "mpc_batch_set_solver: MPC_SOLVER_OSQP (0) or MPC_SOLVER_EXACT (1)");
  b->exact = solver == MPC_SOLVER_EXACT;
  return MPC_OK;
}

int mpc_batch_set_max_iter(mpc_batch *b, int max_iter) {
  if (!b || max_iter <= 0 || max_iter % kCheck != 0) return fail(MPC_E_ARG, "mpc_batch_set_max_iter: a positive multiple of 25 (OSQP's check_termination interval)");
  b->max_iter = max_iter;
  return MPC_OK;
}

int mpc_batch_solve_f64(mpc_batch *b, const double *d_in, double *d_forces, int *d_info, void *stream) {
  if (!b || !d_in || !d_forces) return fail(MPC_E_ARG, "mpc_batch_solve_f64: bad argument");
  return launch_solver(b, nullptr, d_forces, d_info ? d_info : b->d_info, nullptr, reinterpret_cast<hipStream_t>(stream), d_in);
}

int mpc_batch_reset(mpc_batch *b, const int *ids, int k, void *stream) {
  if (!b) return fail(MPC_E_ARG, "mpc_batch_reset: bad argument");
  DeviceGuard guard_(b->device);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (!ids) {
    HIP_TRY(hipMemsetAsync(b->d_state, 0, sizeof(double) * (size_t)b->n * b->state_len, st));
    return MPC_OK;
  }
  if (k <= 0) return MPC_OK;
  int *d_ids = nullptr;
  HIP_TRY(hipMallocAsync(reinterpret_cast<void **>(&d_ids), sizeof(int) * k, st));
  HIP_TRY(hipMemcpyAsync(d_ids, ids, sizeof(int) * k, hipMemcpyHostToDevice, st));
  hipLaunchKernelGGL(reset_kernel, dim3(k), dim3(256), 0, st, b->d_state, b->state_len, d_ids, k, b->n);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipFreeAsync(d_ids, st));
  return MPC_OK;
}

int mpc_batch_reset_device(mpc_batch *b, const int *d_ids, int k, void *stream) {
  if (!b || !d_ids || k < 0) return fail(MPC_E_ARG, "mpc_batch_reset_device: bad argument");
  DeviceGuard guard_(b->device);
  if (k == 0) return MPC_OK;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(reset_kernel, dim3(k), dim3(256), 0, st, b->d_state, b->state_len, d_ids, k, b->n);
  HIP_TRY(hipGetLastError());
  return MPC_OK;
}

int mpc_batch_solve_host(mpc_batch *b, const float *h_in, double *h_forces, int *h_info) {
  if (!b || !h_in || !h_forces) return fail(MPC_E_ARG, "mpc_batch_solve_host: bad argument");
  DeviceGuard guard_(b->device);
  const size_t inlen = 56 + 4 * (size_t)b->h, N = 12 * (size_t)b->h;
  if (!b->d_host_in) {   // staging buffers of the host-pointer entry point, kept for the life of the handle
    HIP_TRY(hipMalloc(&b->d_host_in, sizeof(float) * b->n * inlen));
    if (!b->d_host_f) HIP_TRY(hipMalloc(&b->d_host_f, sizeof(double) * b->n * N));
  }
  HIP_TRY(hipMemcpy(b->d_host_in, h_in, sizeof(float) * b->n * inlen, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(b->d_host_f, h_forces, sizeof(double) * b->n * N, hipMemcpyHostToDevice));   // rows of unsolved robots stay as passed in
  const int rc = mpc_batch_solve(b, b->d_host_in, b->d_host_f, nullptr, nullptr);
  if (rc != MPC_OK) return rc;
  HIP_TRY(hipMemcpy(h_forces, b->d_host_f, sizeof(double) * b->n * N, hipMemcpyDeviceToHost));     // (synchronises with the null stream)
  if (h_info) HIP_TRY(hipMemcpy(h_info, b->d_info, sizeof(int) * b->n * kInfoLen, hipMemcpyDeviceToHost));
  return MPC_OK;
}

int mpc_batch_solve_host_f64(mpc_batch *b, const double *h_in, double *h_forces, int *h_info) {
  if (!b || !h_in || !h_forces) return fail(MPC_E_ARG, "mpc_batch_solve_host_f64: bad argument");
  DeviceGuard guard_(b->device);
  const size_t inlen = 56 + 4 * (size_t)b->h, N = 12 * (size_t)b->h;
  if (!b->d_host_in64) {
    HIP_TRY(hipMalloc(&b->d_host_in64, sizeof(double) * b->n * inlen));
    if (!b->d_host_f) HIP_TRY(hipMalloc(&b->d_host_f, sizeof(double) * b->n * N));
  }
  HIP_TRY(hipMemcpy(b->d_host_in64, h_in, sizeof(double) * b->n * inlen, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(b->d_host_f, h_forces, sizeof(double) * b->n * N, hipMemcpyHostToDevice));   // rows of unsolved robots stay as passed in
  const int rc = mpc_batch_solve_f64(b, b->d_host_in64, b->d_host_f, nullptr, nullptr);
  if (rc != MPC_OK) return rc;
  HIP_TRY(hipMemcpy(h_forces, b->d_host_f, sizeof(double) * b->n * N, hipMemcpyDeviceToHost));     // (synchronises with the null stream)
  if (h_info) HIP_TRY(hipMemcpy(h_info, b->d_info, sizeof(int) * b->n * kInfoLen, hipMemcpyDeviceToHost));
  return MPC_OK;
}

int mpc_batch_enable_timing(mpc_batch *b) {
  if (!b) return fail(MPC_E_ARG, "mpc_batch_enable_timing: bad argument");
  if (!b->timing) {
    for (auto &e3 : b->ev) for (auto &e : e3) HIP_TRY(hipEventCreate(&e));
    b->timing = true;
    b->launches = 0;
  }
  return MPC_OK;
}
int mpc_batch_kernel_times(mpc_batch *b, int last_k, float *ms_assemble, float *ms_solve) {
  if (!b || !b->timing || last_k <= 0 || last_k > kTimingRing || last_k > b->launches || !ms_assemble || !ms_solve)
    return fail(MPC_E_ARG, "mpc_batch_kernel_times: bad argument (enable timing first; at most 64 launches back)");
  DeviceGuard guard_(b->device);
  HIP_TRY(hipDeviceSynchronize());
  for (int i = 0; i < last_k; ++i) {
    hipEvent_t *e = b->ev[(b->launches - last_k + i) % kTimingRing];
    HIP_TRY(hipEventElapsedTime(ms_assemble + i, e[0], e[1]));
    HIP_TRY(hipEventElapsedTime(ms_solve + i, e[1], e[2]));
  }
  return MPC_OK;
}
int mpc_batch_size(const mpc_batch *b) { return b ? b->n : 0; }
int mpc_batch_horizon(const mpc_batch *b) { return b ? b->h : 0; }
long long mpc_batch_device_bytes(const mpc_batch *b) { return b ? b->bytes : 0; }
int mpc_batch_state_len(const mpc_batch *b) { return b ? b->state_len : 0; }
int mpc_batch_get_state(mpc_batch *b, double *h_state) {
  if (!b || !h_state) return fail(MPC_E_ARG, "mpc_batch_get_state: bad argument");
  DeviceGuard guard_(b->device);
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpy(h_state, b->d_state, sizeof(double) * (size_t)b->n * b->state_len, hipMemcpyDeviceToHost));
  return MPC_OK;
}
// Test / debugging access to what the prep kernel handed to the solve kernel in the last launch
int mpc_batch_qp_len(const mpc_batch *b) { return b ? (int)xqp_len_of(b->h) : 0; }
int mpc_batch_scale_len(const mpc_batch *b) { return b ? (int)xsc_len_of(b->h) : 0; }
int mpc_batch_get_qp(mpc_batch *b, double *h_qp) {
  if (!b || !h_qp) return fail(MPC_E_ARG, "mpc_batch_get_qp: bad argument");
  DeviceGuard guard_(b->device);
  return fetch_records(b, h_qp, nullptr);
}
int mpc_batch_get_scale(mpc_batch *b, double *h_sc) {
  if (!b || !h_sc) return fail(MPC_E_ARG, "mpc_batch_get_scale: bad argument");
  DeviceGuard guard_(b->device);
  return fetch_records(b, nullptr, h_sc);
}
int mpc_batch_get_profile(mpc_batch *b, long long *h_prof) {
  if (!b || !h_prof) return fail(MPC_E_ARG, "mpc_batch_get_profile: bad argument");
  DeviceGuard guard_(b->device);
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpy(h_prof, b->d_prof, sizeof(long long) * (size_t)b->n * kProfLen, hipMemcpyDeviceToHost));
  return MPC_OK;
}
int mpc_batch_set_state(mpc_batch *b, const double *h_state) {
  if (!b || !h_state) return fail(MPC_E_ARG, "mpc_batch_set_state: bad argument");
  DeviceGuard guard_(b->device);
  HIP_TRY(hipMemcpy(b->d_state, h_state, sizeof(double) * (size_t)b->n * b->state_len, hipMemcpyHostToDevice));
  return MPC_OK;
}

}  // extern "C"

// ------------------------------------------------------------------------------------------------
// Per-tick controller (controller.h): one thread per robot before and after the solve.
// ------------------------------------------------------------------------------------------------
namespace {

constexpr int kCtrlThreads = 64;   // one wave per workgroup: 4096 robots spread over 64 CUs, and the register cap is 512 (no spills)

__global__ __launch_bounds__(kCtrlThreads) void ctrl_init_kernel(int n, CtrlState *st, const RobotConst *rc, const int *robot_type, const int *gait_id) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r < n) ctrl_init(st[r], rc[robot_type[r]], robot_type[r], gait_id[r]);
}
__global__ __launch_bounds__(kCtrlThreads) void ctrl_reset_kernel(int n, CtrlState *st, const RobotConst *rc, const int *ids, int k) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= k) return;
  const int r = ids ? ids[i] : i;
  if (r >= 0 && r < n) ctrl_reset(st[r], rc[st[r].robot_type]);
}
__global__ void ctrl_set_gait_kernel(int n, CtrlState *st, const int *gait_id) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r < n) st[r].gait_id = gait_id[r];
}
// The controller's two halves of a tick with ONE LANE PER LEG (four lanes per robot, a hardware quad): a lane works on a private copy of
// the robot's state, does its own leg's part (controller.h ctrl_pre_legs / ctrl_pre_rest / ctrl_post with the leg range [leg, leg + 1)) and
// writes back its leg's fields; what concerns the whole robot every lane computes alike and lane 0 writes.  One lane per robot made
// the tick's 24 double-precision sines and cosines and its float16-emulating arithmetic one dependent chain (estimator + pre + post:
// 43 us per tick at 4096 robots, 10 % of a tick).
static_assert(sizeof(CtrlState) == 888, "CtrlState changed: every field must be stored by store_leg_fields or store_robot_fields");
__device__ __forceinline__ void store_leg_fields(CtrlState &d, const CtrlState &s, int leg) {
  d.first_swing[leg] = s.first_swing[leg];
  d.swing_time_remaining[leg] = s.swing_time_remaining[leg];
  d.swing_times[leg] = s.swing_times[leg];
  d.contact_phase[leg] = s.contact_phase[leg];
  d.contact_states[leg] = s.contact_states[leg];
  d.swing_states[leg] = s.swing_states[leg];
  for (int c = 3 * leg; c < 3 * leg + 3; ++c) {
    d.f_ff[c] = s.f_ff[c]; d.p0[c] = s.p0[c]; d.pf[c] = s.pf[c]; d.tp[c] = s.tp[c]; d.tv[c] = s.tv[c]; d.hist[c] = s.hist[c];
    d.q[c] = s.q[c]; d.qd[c] = s.qd[c]; d.p[c] = s.p[c]; d.v[c] = s.v[c]; d.foot_positions[c] = s.foot_positions[c]; d.pfoot[c] = s.pfoot[c];
  }
  for (int c = 9 * leg; c < 9 * leg + 9; ++c) d.J[c] = s.J[c];
}
__device__ __forceinline__ void store_robot_fields(CtrlState &d, const CtrlState &s) {
  d.iter = s.iter; d.first_run = s.first_run; d.pos_z = s.pos_z; d.posz_tick = s.posz_tick; d.do_solve = s.do_solve;
  for (int c = 0; c < 3; ++c) { d.normal[c] = s.normal[c]; d.vbody[c] = s.vbody[c]; }
}
__global__ __launch_bounds__(kCtrlThreads) void ctrl_pre_kernel(int n, CtrlState *st, const RobotConst *rc, GaitTable gt, CtrlParams cp, const float *dof,
                                const float *est, const float *cmd, float *rec, int *active) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x, r = t >> 2, leg = t & 3;
  if (r >= n) return;           // (n robots = 4 n threads: whole quads leave together)
  CtrlState s = st[r];
  const RobotConst &k = rc[s.robot_type];
  ctrl_pre_legs(s, k, dof + (size_t)r * 24, leg, leg + 1);
  // every lane needs the foot positions of all four legs (centre-of-mass height, ground-normal fit, the solver record)
  const int q0 = (int)(threadIdx.x & 63u) & ~3;
  float fp[12];
#pragma unroll
  for (int l = 0; l < 4; ++l)
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float mine = c == 0 ? s.foot_positions[3 * leg] : (c == 1 ? s.foot_positions[3 * leg + 1] : s.foot_positions[3 * leg + 2]);
      fp[3 * l + c] = __shfl(mine, q0 + l, 64);
    }
#pragma unroll
  for (int c = 0; c < 12; ++c) s.foot_positions[c] = fp[c];
  ctrl_pre_rest(s, k, gt, cp, est + (size_t)r * kEstLen, cmd + (size_t)r * 16, rec + (size_t)r * (56 + 4 * cp.horizon), leg, leg + 1, leg == 0);
  store_leg_fields(st[r], s, leg);
  if (leg == 0) {
    store_robot_fields(st[r], s);
    active[r] = s.do_solve;
  }
}
__global__ __launch_bounds__(kCtrlThreads) void estimator_kernel(int n, const CtrlState *st, const float *body, float *est) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n) return;
  const float nrm[3] = {st[r].normal[0], st[r].normal[1], st[r].normal[2]};
  float e[kEstLen];
  estimator_update(body + (size_t)r * 13, nrm, e);
  for (int k = 0; k < kEstLen; ++k) est[(size_t)r * kEstLen + k] = e[k];
}
__global__ __launch_bounds__(kCtrlThreads) void fsm_init_kernel(int n, CtrlState *st, FsmState *fs, const RobotConst *rc, const int *mode, int op_mode, const int *ids, int k, int fresh,
                                                                double *solver_state, int state_len) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= k) return;
  const int r = ids ? ids[i] : i;
  if (r < 0 || r >= n) return;
  CtrlState s = st[r];
  FsmState f = fs[r];
  if (fresh) fsm_init(f, mode[r], op_mode, s, rc[s.robot_type], 0.f);
  else fsm_reinit(f, mode[r], op_mode, s, rc[s.robot_type], f.last_rb22);
  // entering LOCOMOTION runs cMPC.initialize (FSM_State_Locomotion.py:32-42 -> ConvexMPCLocomotion.py:102-108): a NEW ConvexMpc object,
  // i.e. x = y = z = 0, rho = 0.1 and an "osqp_setup" first call.  (fsm_tick clears entered_loco at its top, so it is consumed here.)
  if (f.entered_loco && solver_state)
    for (int q = 0; q < state_len; ++q) solver_state[(size_t)r * state_len + q] = 0.0;
  st[r] = s; fs[r] = f;
}
// RobotRunnerFSM.run up to the solver launch: fsm_tick, then ctrl_pre for the robots whose state runs the locomotion controller
__global__ __launch_bounds__(kCtrlThreads) void fsm_pre_kernel(int n, CtrlState *st, FsmState *fs, const RobotConst *rc, GaitTable gt, CtrlParams cp, FsmParams fp,
                               const float *dof, const float *body, const float *est, const float *cmd, const int *request, float *rec,
                               int *active, double *solver_state, int state_len) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n) return;
  CtrlState s = st[r];
  FsmState f = fs[r];
  fsm_tick(f, s, rc[s.robot_type], fp, dof + (size_t)r * 24, body + (size_t)r * 13, request[r]);
  if (f.entered_loco)    // cMPC.initialize built a new ConvexMpc (ConvexMPCLocomotion.py:102-108): the next solve is a cold one
    for (int k = 0; k < state_len; ++k) solver_state[(size_t)r * state_len + k] = 0.0;
  int act = 0;
  if (f.run_loco) {
    ctrl_pre(s, rc[s.robot_type], gt, cp, dof + (size_t)r * 24, est + (size_t)r * kEstLen, cmd + (size_t)r * 16, rec + (size_t)r * (56 + 4 * cp.horizon));
    act = s.do_solve;
  }
  active[r] = act;
  st[r] = s; fs[r] = f;
}
__global__ __launch_bounds__(kCtrlThreads) void fsm_post_kernel(int n, CtrlState *st, const FsmState *fs, const RobotConst *rc, int horizon, const double *forces, const int *info,
                                float *torques) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n) return;
  if (fs[r].run_loco) {
    CtrlState s = st[r];
    ctrl_post(s, rc[s.robot_type], forces + (size_t)r * 12 * horizon, info[(size_t)r * kInfoLen + 1] == kStSolved, torques + (size_t)r * 12);
    st[r] = s;
  } else {
    fsm_joint_torques(fs[r], st[r], torques + (size_t)r * 12);
  }
}
__global__ __launch_bounds__(kCtrlThreads) void ctrl_post_kernel(int n, CtrlState *st, const RobotConst *rc, int horizon, const double *forces, const int *info,
                                 float *torques) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x, r = t >> 2, leg = t & 3;      // one lane per leg (see ctrl_pre_kernel)
  if (r >= n) return;
  CtrlState s = st[r];
  ctrl_post(s, rc[s.robot_type], forces + (size_t)r * 12 * horizon, info[(size_t)r * kInfoLen + 1] == kStSolved, torques + (size_t)r * 12, leg, leg + 1);
  store_leg_fields(st[r], s, leg);
}

}  // namespace

struct mpc_ctrl {
  int n = 0;
  mpc_batch *solver = nullptr;
  CtrlState *d_state = nullptr;
  RobotConst *d_rc = nullptr;
  int *d_robot_type = nullptr, *d_gait = nullptr, *d_active = nullptr, *d_info = nullptr;
  float *d_rec = nullptr, *d_est = nullptr;
  double *d_forces = nullptr;
  GaitTable gt;
  CtrlParams cp;
  std::vector<int> h_iter;        // host mirror of every robot's iterationCounter (deterministic outside the FSM): lets a tick
  bool mirror_valid = true;       // on which no robot is due for an MPC update skip the solver launches
  FsmState *d_fsm = nullptr;      // control FSM (allocated by mpc_ctrl_fsm_init)
  int *d_fsm_mode = nullptr;      // per-robot control mode of the last (re)initialisation
  FsmParams fp{};
  int fsm_op_mode = kOpNormal;
};

extern "C" {

void mpc_ctrl_destroy(mpc_ctrl *c) {
  if (!c) return;
  mpc_batch_destroy(c->solver);
  void *ptrs[] = {c->d_state, c->d_rc, c->d_robot_type, c->d_gait, c->d_active, c->d_info, c->d_rec, c->d_est, c->d_forces, c->d_fsm, c->d_fsm_mode};
  for (void *p : ptrs) if (p) (void)hipFree(p);
  delete c;
}

int mpc_ctrl_create(mpc_ctrl **out, int n, int horizon, double controller_dt, int iters_between_mpc, double alpha, int flat_ground,
                    const int *robot_type, const int *gait_id, int n_types, const double *robot_table, const int *gait_off,
                    const int *gait_dur) {
  if (!out || n <= 0 || !robot_type || !gait_id || n_types <= 0 || !robot_table || !gait_off || !gait_dur || iters_between_mpc <= 0)
    return fail(MPC_E_ARG, "mpc_ctrl_create: bad argument");
  std::vector<double> mass(n), inertia((size_t)n * 9, 0.0);
  for (int r = 0; r < n; ++r) {
    if (robot_type[r] < 0 || robot_type[r] >= n_types || gait_id[r] < 0 || gait_id[r] >= kNumGaitIds) return fail(MPC_E_ARG, "mpc_ctrl_create: robot_type / gait_id out of range");
    const double *row = robot_table + 25 * robot_type[r];
    mass[r] = row[6];
    inertia[9 * (size_t)r] = row[7]; inertia[9 * (size_t)r + 4] = row[8]; inertia[9 * (size_t)r + 8] = row[9];
  }
  mpc_ctrl *c = new mpc_ctrl();
  c->n = n;
  const double dt_mpc = controller_dt * iters_between_mpc;
  int rc0 = mpc_batch_create(&c->solver, n, horizon, dt_mpc, alpha, mass.data(), inertia.data());
  if (rc0 != MPC_OK) { delete c; return rc0; }
  c->cp = CtrlParams{controller_dt, iters_between_mpc, dt_mpc, horizon, flat_ground};
  c->gt.n_seg = horizon;
  for (int g = 0; g < kNumGaitIds; ++g)
    for (int j = 0; j < 4; ++j) { c->gt.offsets[g][j] = (float)gait_off[4 * g + j]; c->gt.durations[g][j] = (float)gait_dur[4 * g + j]; }
  std::vector<RobotConst> rcs(n_types);
  for (int t = 0; t < n_types; ++t) {
    const double *row = robot_table + 25 * t;
    rcs[t].abad = row[0]; rcs[t].hip = row[1]; rcs[t].knee = row[2];
    for (int k = 0; k < 3; ++k) rcs[t].hiploc[k] = (float)row[3 + k];
    rcs[t].body_height = row[10]; rcs[t].mu = (float)row[11];
    for (int k = 0; k < 13; ++k) rcs[t].weights[k] = (float)row[12 + k];
  }
  const size_t inlen = 56 + 4 * (size_t)horizon;
  hipError_t e;
  if ((e = hipMalloc(&c->d_state, sizeof(CtrlState) * n)) != hipSuccess || (e = hipMalloc(&c->d_rc, sizeof(RobotConst) * n_types)) != hipSuccess ||
      (e = hipMalloc(&c->d_robot_type, sizeof(int) * n)) != hipSuccess || (e = hipMalloc(&c->d_gait, sizeof(int) * n)) != hipSuccess ||
      (e = hipMalloc(&c->d_active, sizeof(int) * n)) != hipSuccess || (e = hipMalloc(&c->d_info, sizeof(int) * (size_t)n * kInfoLen)) != hipSuccess ||
      (e = hipMalloc(&c->d_rec, sizeof(float) * n * inlen)) != hipSuccess || (e = hipMalloc(&c->d_est, sizeof(float) * (size_t)n * kEstLen)) != hipSuccess || (e = hipMalloc(&c->d_forces, sizeof(double) * (size_t)n * 12 * horizon)) != hipSuccess ||
      (e = hipMemcpy(c->d_rc, rcs.data(), sizeof(RobotConst) * n_types, hipMemcpyHostToDevice)) != hipSuccess ||
      (e = hipMemcpy(c->d_robot_type, robot_type, sizeof(int) * n, hipMemcpyHostToDevice)) != hipSuccess ||
      (e = hipMemcpy(c->d_gait, gait_id, sizeof(int) * n, hipMemcpyHostToDevice)) != hipSuccess ||
      (e = hipMemset(c->d_info, 0, sizeof(int) * (size_t)n * kInfoLen)) != hipSuccess ||
      (e = hipMemset(c->d_rec, 0, sizeof(float) * n * inlen)) != hipSuccess ||
      (e = hipMemset(c->d_forces, 0, sizeof(double) * (size_t)n * 12 * horizon)) != hipSuccess) {
    mpc_ctrl_destroy(c);
    return fail(MPC_E_HIP, std::string("mpc_ctrl_create: ") + hipGetErrorString(e));
  }
  hipLaunchKernelGGL(ctrl_init_kernel, dim3((n + kCtrlThreads - 1) / kCtrlThreads), dim3(kCtrlThreads), 0, nullptr, n, c->d_state, c->d_rc, c->d_robot_type, c->d_gait);
  if ((e = hipDeviceSynchronize()) != hipSuccess) { mpc_ctrl_destroy(c); return fail(MPC_E_HIP, std::string("mpc_ctrl_create: ") + hipGetErrorString(e)); }
  *out = c;
  return MPC_OK;
}

int mpc_ctrl_step(mpc_ctrl *c, const float *d_dof, const float *d_est, const float *d_cmd, float *d_torques, void *stream) {
  if (!c || !d_dof || !d_est || !d_cmd || !d_torques) return fail(MPC_E_ARG, "mpc_ctrl_step: bad argument");
  DeviceGuard guard_(c->solver->device);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int n = c->n, blocks4 = (4 * n + kCtrlThreads - 1) / kCtrlThreads;      // (ctrl_pre / ctrl_post: one lane per leg)
  bool any_due = true;
  if (c->mirror_valid) {   // ConvexMPCLocomotion.run: iterationCounter += 1, MPC update when it is a multiple of iterationsBetweenMPC
    if ((int)c->h_iter.size() != n) c->h_iter.assign(n, 0);
    any_due = false;
    for (int r = 0; r < n; ++r) any_due |= (++c->h_iter[r] % c->cp.iters_between_mpc) == 0;
  }
  hipLaunchKernelGGL(ctrl_pre_kernel, dim3(blocks4), dim3(kCtrlThreads), 0, st, n, c->d_state, c->d_rc, c->gt, c->cp, d_dof, d_est, d_cmd, c->d_rec, c->d_active);
  HIP_TRY(hipGetLastError());
  mpc_batch *b = c->solver;
  if (any_due) {
    int rc = launch_solver(b, c->d_rec, c->d_forces, c->d_info, c->d_active, st);
    if (rc != MPC_OK) return rc;
  }
  hipLaunchKernelGGL(ctrl_post_kernel, dim3(blocks4), dim3(kCtrlThreads), 0, st, n, c->d_state, c->d_rc, c->cp.horizon, c->d_forces, c->d_info, d_torques);
  HIP_TRY(hipGetLastError());
  return MPC_OK;
}

int mpc_ctrl_run(mpc_ctrl *c, const float *d_dof, const float *d_body, const float *d_cmd, float *d_torques, void *stream) {
  if (!c || !d_dof || !d_body || !d_cmd || !d_torques) return fail(MPC_E_ARG, "mpc_ctrl_run: bad argument");
  DeviceGuard guard_(c->solver->device);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(estimator_kernel, dim3((c->n + kCtrlThreads - 1) / kCtrlThreads), dim3(kCtrlThreads), 0, st, c->n, c->d_state, d_body, c->d_est);
  HIP_TRY(hipGetLastError());
  return mpc_ctrl_step(c, d_dof, c->d_est, d_cmd, d_torques, stream);
}

int mpc_ctrl_reset(mpc_ctrl *c, const int *ids, int k, void *stream) {
  if (!c) return fail(MPC_E_ARG, "mpc_ctrl_reset: bad argument");
  DeviceGuard guard_(c->solver->device);
  if (!ids) c->h_iter.assign(c->n, 0);
  else for (int i = 0; i < k; ++i) if (ids[i] >= 0 && ids[i] < c->n && (int)c->h_iter.size() == c->n) c->h_iter[ids[i]] = 0;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  int rc = mpc_batch_reset(c->solver, ids, k, stream);   // new ConvexMpc object = cold solver (ConvexMPCLocomotion.py:102-108)
  if (rc != MPC_OK) return rc;
  if (!ids) {
    hipLaunchKernelGGL(ctrl_reset_kernel, dim3((c->n + kCtrlThreads - 1) / kCtrlThreads), dim3(kCtrlThreads), 0, st, c->n, c->d_state, c->d_rc, (const int *)nullptr, c->n);
    HIP_TRY(hipGetLastError());
    return MPC_OK;
  }
  if (k <= 0) return MPC_OK;
  int *d_ids = nullptr;
  HIP_TRY(hipMallocAsync(reinterpret_cast<void **>(&d_ids), sizeof(int) * k, st));
  HIP_TRY(hipMemcpyAsync(d_ids, ids, sizeof(int) * k, hipMemcpyHostToDevice, st));
  hipLaunchKernelGGL(ctrl_reset_kernel, dim3((k + kCtrlThreads - 1) / kCtrlThreads), dim3(kCtrlThreads), 0, st, c->n, c->d_state, c->d_rc, d_ids, k);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipFreeAsync(d_ids, st));
  return MPC_OK;
}

int mpc_ctrl_reset_device(mpc_ctrl *c, const int *d_ids, int k, void *stream) {
  if (!c || !d_ids || k < 0) return fail(MPC_E_ARG, "mpc_ctrl_reset_device: bad argument");
  DeviceGuard guard_(c->solver->device);
  if (k == 0) return MPC_OK;
  c->mirror_valid = false;     // the host copy of the MPC counters cannot follow ids it never sees: launch the solver on every tick (active mask)
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  int rc = mpc_batch_reset_device(c->solver, d_ids, k, stream);
  if (rc != MPC_OK) return rc;
  hipLaunchKernelGGL(ctrl_reset_kernel, dim3((k + kCtrlThreads - 1) / kCtrlThreads), dim3(kCtrlThreads), 0, st, c->n, c->d_state, c->d_rc, d_ids, k);
  HIP_TRY(hipGetLastError());
  return MPC_OK;
}

int mpc_ctrl_set_solver(mpc_ctrl *c, int solver) {
  if (!c) return fail(MPC_E_ARG, "mpc_ctrl_set_solver: bad argument");
  return mpc_batch_set_solver(c->solver, solver);
}

int mpc_ctrl_set_gait(mpc_ctrl *c, const int *gait_id, void *stream) {
  if (!c || !gait_id) return fail(MPC_E_ARG, "mpc_ctrl_set_gait: bad argument");
  DeviceGuard guard_(c->solver->device);
  for (int r = 0; r < c->n; ++r) if (gait_id[r] < 0 || gait_id[r] >= kNumGaitIds) return fail(MPC_E_ARG, "mpc_ctrl_set_gait: gait id out of range");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  HIP_TRY(hipMemcpyAsync(c->d_gait, gait_id, sizeof(int) * c->n, hipMemcpyHostToDevice, st));
  hipLaunchKernelGGL(ctrl_set_gait_kernel, dim3((c->n + kCtrlThreads - 1) / kCtrlThreads), dim3(kCtrlThreads), 0, st, c->n, c->d_state, c->d_gait);
  HIP_TRY(hipGetLastError());
  return MPC_OK;
}

int mpc_ctrl_fsm_init(mpc_ctrl *c, const int *control_mode, int operating_mode, int check_safety, void *stream) {
  if (!c || !control_mode || (operating_mode != kOpTest && operating_mode != kOpNormal)) return fail(MPC_E_ARG, "mpc_ctrl_fsm_init: bad argument");
  DeviceGuard guard_(c->solver->device);
  for (int r = 0; r < c->n; ++r)
    if (control_mode[r] != kFsmPassive && control_mode[r] != kFsmLocomotion && control_mode[r] != kFsmRecoveryStand)
      return fail(MPC_E_ARG, "mpc_ctrl_fsm_init: control mode must be 0 (PASSIVE), 4 (LOCOMOTION) or 6 (RECOVERY_STAND)");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (!c->d_fsm) {
    HIP_TRY(hipMalloc(&c->d_fsm, sizeof(FsmState) * c->n));
    HIP_TRY(hipMalloc(&c->d_fsm_mode, sizeof(int) * c->n));
  }
  c->mirror_valid = false;       // from here on the device decides which robots run the locomotion controller
  c->fp = fsm_params(c->cp.dt, check_safety);
  c->fsm_op_mode = operating_mode;
  HIP_TRY(hipMemcpyAsync(c->d_fsm_mode, control_mode, sizeof(int) * c->n, hipMemcpyHostToDevice, st));
  int rc = mpc_batch_reset(c->solver, nullptr, 0, stream);   // RobotRunnerFSM.init builds fresh objects
  if (rc != MPC_OK) return rc;
  hipLaunchKernelGGL(ctrl_init_kernel, dim3((c->n + kCtrlThreads - 1) / kCtrlThreads), dim3(kCtrlThreads), 0, st, c->n, c->d_state, c->d_rc, c->d_robot_type, c->d_gait);
  hipLaunchKernelGGL(fsm_init_kernel, dim3((c->n + kCtrlThreads - 1) / kCtrlThreads), dim3(kCtrlThreads), 0, st, c->n, c->d_state, c->d_fsm, c->d_rc, c->d_fsm_mode, operating_mode,
                     (const int *)nullptr, c->n, 1, (double *)nullptr, 0);   // (the whole solver was reset above)
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(st));                          // control_mode is a host buffer
  return MPC_OK;
}

int mpc_ctrl_fsm_reset(mpc_ctrl *c, const int *ids, int k, const int *control_mode, void *stream) {
  if (!c || !c->d_fsm || (ids && k < 0)) return fail(MPC_E_ARG, "mpc_ctrl_fsm_reset: bad argument (mpc_ctrl_fsm_init first)");
  DeviceGuard guard_(c->solver->device);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (control_mode) {   // [n] entries, like mpc_ctrl_fsm_init
    for (int r = 0; r < c->n; ++r)
      if (control_mode[r] != kFsmPassive && control_mode[r] != kFsmLocomotion && control_mode[r] != kFsmRecoveryStand)
        return fail(MPC_E_ARG, "mpc_ctrl_fsm_reset: control mode must be 0 (PASSIVE), 4 (LOCOMOTION) or 6 (RECOVERY_STAND)");
    HIP_TRY(hipMemcpyAsync(c->d_fsm_mode, control_mode, sizeof(int) * c->n, hipMemcpyHostToDevice, st));
  }
  const int cnt = ids ? k : c->n;
  if (cnt == 0) return MPC_OK;
  int *d_ids = nullptr;
  if (ids) {
    HIP_TRY(hipMallocAsync(reinterpret_cast<void **>(&d_ids), sizeof(int) * k, st));
    HIP_TRY(hipMemcpyAsync(d_ids, ids, sizeof(int) * k, hipMemcpyHostToDevice, st));
  }
  hipLaunchKernelGGL(fsm_init_kernel, dim3((cnt + kCtrlThreads - 1) / kCtrlThreads), dim3(kCtrlThreads), 0, st, c->n, c->d_state, c->d_fsm, c->d_rc, c->d_fsm_mode, c->fsm_op_mode, d_ids, cnt, 0,
                     c->solver->d_state, c->solver->state_len);
  HIP_TRY(hipGetLastError());
  if (d_ids) HIP_TRY(hipFreeAsync(d_ids, st));
  HIP_TRY(hipStreamSynchronize(st));
  return MPC_OK;
}

int mpc_ctrl_run_fsm(mpc_ctrl *c, const float *d_dof, const float *d_body, const float *d_cmd, const int *d_request, float *d_torques, void *stream) {
  if (!c || !d_dof || !d_body || !d_cmd || !d_request || !d_torques) return fail(MPC_E_ARG, "mpc_ctrl_run_fsm: bad argument");
  DeviceGuard guard_(c->solver->device);
  if (!c->d_fsm) return fail(MPC_E_ARG, "mpc_ctrl_run_fsm: call mpc_ctrl_fsm_init first");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int n = c->n, blocks = (n + kCtrlThreads - 1) / kCtrlThreads;
  mpc_batch *b = c->solver;
  hipLaunchKernelGGL(estimator_kernel, dim3(blocks), dim3(kCtrlThreads), 0, st, n, c->d_state, d_body, c->d_est);
  hipLaunchKernelGGL(fsm_pre_kernel, dim3(blocks), dim3(kCtrlThreads), 0, st, n, c->d_state, c->d_fsm, c->d_rc, c->gt, c->cp, c->fp, d_dof, d_body, c->d_est, d_cmd,
                     d_request, c->d_rec, c->d_active, b->d_state, b->state_len);
  HIP_TRY(hipGetLastError());
  int rc = launch_solver(b, c->d_rec, c->d_forces, c->d_info, c->d_active, st);
  if (rc != MPC_OK) return rc;
  hipLaunchKernelGGL(fsm_post_kernel, dim3(blocks), dim3(kCtrlThreads), 0, st, n, c->d_state, c->d_fsm, c->d_rc, c->cp.horizon, c->d_forces, c->d_info, d_torques);
  HIP_TRY(hipGetLastError());
  return MPC_OK;
}

int mpc_ctrl_fsm_state(mpc_ctrl *c, int *h_out) {
  if (!c || !c->d_fsm || !h_out) return fail(MPC_E_ARG, "mpc_ctrl_fsm_state: bad argument");
  DeviceGuard guard_(c->solver->device);
  std::vector<FsmState> h(c->n);
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpy(h.data(), c->d_fsm, sizeof(FsmState) * c->n, hipMemcpyDeviceToHost));
  for (int r = 0; r < c->n; ++r) { h_out[4 * r] = h[r].cur; h_out[4 * r + 1] = h[r].op_mode; h_out[4 * r + 2] = h[r].rs_flag; h_out[4 * r + 3] = h[r].unsafe; }
  return MPC_OK;
}

int mpc_ctrl_solver_info(mpc_ctrl *c, int *h_info) {
  if (!c || !h_info) return fail(MPC_E_ARG, "mpc_ctrl_solver_info: bad argument");
  DeviceGuard guard_(c->solver->device);
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpy(h_info, c->d_info, sizeof(int) * (size_t)c->n * kInfoLen, hipMemcpyDeviceToHost));
  return MPC_OK;
}

// ---- weight policy (RL_Environment/WeightPolicy.py) ------------------------------------------------------
struct mpc_policy {
  policy::Net net{};
  float *d_params = nullptr;   // all weights and biases, one allocation
  size_t lds = 0;
};

namespace {
__global__ void ctrl_estimate_kernel(int n, const CtrlState *st, const float *est_in, float *est_out, float *normal_out) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n) return;
  if (est_out) for (int k = 0; k < 18; ++k) est_out[18 * r + k] = est_in[18 * r + k];
  if (normal_out) for (int k = 0; k < 3; ++k) normal_out[3 * r + k] = st[r].normal[k];
}
}  // namespace

void mpc_policy_destroy(mpc_policy *p) {
  if (!p) return;
  if (p->d_params) (void)hipFree(p->d_params);
  delete p;
}

int mpc_policy_create(mpc_policy **out, int n_layers, const int *dims, const float *const *weights, const float *const *biases,
                      const float *act_scale, const float *act_const) {
  if (!out || n_layers <= 0 || n_layers > policy::kMaxLayers || !dims || !weights || !biases || !act_scale || !act_const)
    return fail(MPC_E_ARG, "mpc_policy_create: bad argument");
  size_t total = 0;
  for (int l = 0; l < n_layers; ++l) {
    if (dims[l] <= 0 || dims[l + 1] <= 0 || dims[l] % 8 != 0 || !weights[l] || !biases[l])
      return fail(MPC_E_ARG, "mpc_policy_create: layer input widths must be positive multiples of 8");
    total += (size_t)dims[l] * dims[l + 1] + (((size_t)dims[l + 1] + 7) / 8) * 8;   // keeps every array 32-byte aligned
  }
  if (dims[n_layers] > 16) return fail(MPC_E_ARG, "mpc_policy_create: at most 16 outputs");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail(MPC_E_NODEVICE, "mpc_policy_create: no HIP device");
  mpc_policy *p = new mpc_policy();
  p->net.n_layers = n_layers;
  for (int l = 0; l <= n_layers; ++l) p->net.dims[l] = dims[l];
  std::vector<float> host(total, 0.f);
  if (hipMalloc(&p->d_params, sizeof(float) * total) != hipSuccess) { delete p; return fail(MPC_E_HIP, "mpc_policy_create: hipMalloc"); }
  size_t off = 0;
  for (int l = 0; l < n_layers; ++l) {
    const size_t nw = (size_t)dims[l] * dims[l + 1], nb = (((size_t)dims[l + 1] + 7) / 8) * 8;
    std::memcpy(host.data() + off, weights[l], sizeof(float) * nw);
    p->net.w[l] = p->d_params + off; off += nw;
    std::memcpy(host.data() + off, biases[l], sizeof(float) * dims[l + 1]);
    p->net.b[l] = p->d_params + off; off += nb;
  }
  for (int k = 0; k < dims[n_layers]; ++k) { p->net.scale[k] = act_scale[k]; p->net.shift[k] = act_const[k]; }
  p->lds = policy::lds_bytes(p->net);
  if (p->lds > 160 * 1024) { mpc_policy_destroy(p); return fail(MPC_E_ARG, "mpc_policy_create: layers too wide for one CU's LDS"); }
  if (hipMemcpy(p->d_params, host.data(), sizeof(float) * total, hipMemcpyHostToDevice) != hipSuccess ||
      hipFuncSetAttribute(reinterpret_cast<const void *>(policy::mlp_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)p->lds) != hipSuccess) {
    mpc_policy_destroy(p);
    return fail(MPC_E_HIP, "mpc_policy_create: device set-up failed");
  }
  *out = p;
  return MPC_OK;
}

int mpc_policy_step(mpc_policy *p, int n, const float *d_obs, float *d_actions, float *d_weights, void *stream) {
  if (!p || n <= 0 || !d_obs || !d_weights) return fail(MPC_E_ARG, "mpc_policy_step: bad argument");
  const int blocks = (n + policy::kRows - 1) / policy::kRows;
  hipLaunchKernelGGL(policy::mlp_kernel, dim3(blocks), dim3(policy::kThreads), p->lds, (hipStream_t)stream, p->net, n, d_obs, d_actions, d_weights);
  HIP_TRY(hipGetLastError());
  return MPC_OK;
}

int mpc_policy_observations(int n, const float *d_dof, const float *d_est, const float *d_normal, const float *d_cmd3,
                            const float *d_prev_actions, const float *scales4, float *d_obs, void *stream) {
  if (n <= 0 || !d_dof || !d_est || !d_normal || !d_cmd3 || !d_prev_actions || !scales4 || !d_obs)
    return fail(MPC_E_ARG, "mpc_policy_observations: bad argument");
  hipLaunchKernelGGL(policy::observations_kernel, dim3((n * 48 + 255) / 256), dim3(256), 0, (hipStream_t)stream, n, d_dof, d_est, d_normal,
                     d_cmd3, d_prev_actions, scales4[0], scales4[1], scales4[2], scales4[3], d_obs);
  HIP_TRY(hipGetLastError());
  return MPC_OK;
}

int mpc_pack_commands(int n, const float *d_cmd3, const float *d_weights12, float *d_cmd16, void *stream) {
  if (n <= 0 || !d_cmd3 || !d_weights12 || !d_cmd16) return fail(MPC_E_ARG, "mpc_pack_commands: bad argument");
  hipLaunchKernelGGL(policy::pack_commands_kernel, dim3((n * 16 + 255) / 256), dim3(256), 0, (hipStream_t)stream, n, d_cmd3, d_weights12, d_cmd16);
  HIP_TRY(hipGetLastError());
  return MPC_OK;
}

int mpc_ctrl_update_estimate(mpc_ctrl *c, const float *d_body, void *stream) {
  if (!c || !d_body) return fail(MPC_E_ARG, "mpc_ctrl_update_estimate: bad argument");
  DeviceGuard guard_(c->solver->device);
  hipLaunchKernelGGL(estimator_kernel, dim3((c->n + kCtrlThreads - 1) / kCtrlThreads), dim3(kCtrlThreads), 0, (hipStream_t)stream, c->n, c->d_state, d_body, c->d_est);
  HIP_TRY(hipGetLastError());
  return MPC_OK;
}

int mpc_ctrl_estimate(mpc_ctrl *c, float *d_est, float *d_ground_normal, void *stream) {
  if (!c || (!d_est && !d_ground_normal)) return fail(MPC_E_ARG, "mpc_ctrl_estimate: bad argument");
  DeviceGuard guard_(c->solver->device);
  hipLaunchKernelGGL(ctrl_estimate_kernel, dim3((c->n + kCtrlThreads - 1) / kCtrlThreads), dim3(kCtrlThreads), 0, (hipStream_t)stream, c->n, c->d_state, c->d_est, d_est, d_ground_normal);
  HIP_TRY(hipGetLastError());
  return MPC_OK;
}

}  // extern "C"
