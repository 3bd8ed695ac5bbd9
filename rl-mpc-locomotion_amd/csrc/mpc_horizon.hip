// mpc_horizon.hip -- the gfx950 kernels of ONE planning horizon (-DMPC_H=h) and the HorizonOps entry the library finds them by
// (mpc_horizon.h).  One workgroup per robot and kernel: assembly and Ruiz scaling (mpc_core.h, dense P in register tiles), then the
// OSQP iteration in the wrench space (mpc_wrench.h; one wavefront per robot up to h = 10).  All arithmetic fp64.
#include <hip/hip_runtime.h>

#ifndef MPC_H
#error "compile with -DMPC_H=<planning horizon>"
#endif
#define MPC_LOCKSTEP 1   // a single-wavefront workgroup executes its LDS instructions in program order (mpc_wrench.h: Shared::NBUF)
#include "mpc_horizon.h"
#include "mpc_device.h"
#include "mpc_wrench.h"

using namespace mpc;

// minimum waves per SIMD the register allocator must leave room for in the prep kernel: 3 -> a 256-thread workgroup (h = 10)
// gets 168 VGPRs and three robots share a CU.  Four (128 VGPRs, 88 B of scratch, 36 KB of LDS each): round 2's kernel lost with them
// (0.278 against 0.235 ms per 4096 robots); this round's gains 2 % (0.205 -> 0.201 ms) and pays with 96 MB of spill traffic per launch
// (fabric side 38 -> 133 MB, profiles/r04_ab_prep_h20.txt): not taken.
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


// Solve kernel (mpc_wrench.h Solver): ADMM + polish of every active robot, from the QP and scale records.  EXACT: the
// exact-optimum mode (the reference's qpOASES branch) -- a separate instantiation, so that its outer loop does not touch the
// register allocation of the OSQP mode.
// (EXACT, multi-wave: the full register budget -- this instantiation is the exact mode's rarely used second launch; at two waves per SIMD it
// keeps 2.2 KB of scratch per lane, at one 1.1 KB.)
template <int H, bool EXACT>
__global__ __launch_bounds__(Cfg<H>::TW, (Cfg<H>::TW <= 64 ? MPC_SOLVE_MIN_WAVES : (EXACT ? 1 : MPC_SOLVE_MIN_WAVES_WIDE))) void mpc_solve_kernel(
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
    int *__restrict__ ready, int *__restrict__ seed) {
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
  sv.seedrec = seed ? seed + (size_t)robot * C::NF : nullptr;
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
    // The job's results are device-coherent stores (MPC_GST: sc1, write-through), so they need no L2 write-back -- only to have COMPLETED
    // before the entry is published.  That wait is spelled out: every thread waits for its own stores (s_waitcnt vmcnt(0)), the phase
    // boundary below (barrier / single-wave order) joins the threads, then thread 0 publishes.  The workgroup-scope release fence is
    // there for the COMPILER's ordering only; the memory model does not let it synchronise two workgroups, and nothing may depend on
    // how it happens to be lowered (tests/test_isa_budget.py checks the s_waitcnt on the ISA of every instantiation).
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_s_waitcnt(kWaitVm0);
    ex.par([](WThread<H> &) {});
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
    int n, const RobotModel *__restrict__ models, const float *__restrict__ in, const double *__restrict__ in64, const _Float16 *__restrict__ in16, const double *__restrict__ state,
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
  Assembler<H, Ex> am{ex, sh.as, mdl, in ? in + (size_t)robot * C::IN_LEN : nullptr, in64 ? in64 + (size_t)robot * C::IN_LEN : nullptr, in16 ? in16 + (size_t)robot * C::IN_LEN : nullptr, sh.u12, qpr, prof ? prof + (size_t)robot * kProfLen : nullptr};
  am.run();
  Scaler<H, Ex> sk{ex, sh.sc, state + (size_t)robot * state_len<H>(), sh.u12, mdl.alpha, qpr, sc + (size_t)robot * C::SC_LEN, prof ? prof + (size_t)robot * kProfLen : nullptr};
  sk.run();
}

template <int H>
int launch(const LaunchArgs &a) {
  if (a.ev) (void)hipEventRecord(a.ev[0], a.stream);
  hipLaunchKernelGGL(mpc_prep_kernel<H>, dim3(a.n + 1), dim3(Cfg<H>::T), 0, a.stream, a.n, a.models, a.in, a.in64, a.in16, a.state, a.qp, a.sc, a.prof, a.active, a.order, a.hist,
                     a.hist_slot, a.sched, a.ready);
  if (a.ev) (void)hipEventRecord(a.ev[1], a.stream);
  if (a.exact) {
    hipLaunchKernelGGL((mpc_exact_kernel<H>), dim3(a.n), dim3(Cfg<H>::TW), 0, a.stream, a.models, a.state, a.qp, a.sc, a.forces, a.info, a.prof, a.order, a.sched, a.ready, a.seed);
    hipLaunchKernelGGL((mpc_solve_kernel<H, true>), dim3(a.n), dim3(Cfg<H>::TW), 0, a.stream, a.n, a.models, a.state, a.qp, a.sc, a.forces, a.info, a.prof, a.order, a.sched, a.ready,
                       a.max_iter);
  }
  else if (a.job_slots > 0) {   // persistent workgroups, ADMM and polish as separate jobs (h = 16: 2.68 -> 2.28 ms, h = 20: 3.36 -> 2.78 ms per 4096 robots)
    int slots = Cfg<H>::TW <= 64 ? a.job_slots : a.job_slots / 2;         // (multi-wave workgroups: two per CU)
    if (slots < 1) slots = 1;
    hipLaunchKernelGGL((mpc_solve_jobs_kernel<H>), dim3(a.n < slots ? a.n : slots), dim3(Cfg<H>::TW), 0, a.stream, a.models, a.state, a.qp, a.sc, a.forces, a.info, a.prof, a.order,
                       a.sched, a.ready, a.max_iter);
  }
  else hipLaunchKernelGGL((mpc_solve_kernel<H, false>), dim3(a.n), dim3(Cfg<H>::TW), 0, a.stream, a.n, a.models, a.state, a.qp, a.sc, a.forces, a.info, a.prof, a.order, a.sched,
                          a.ready, a.max_iter);
  if (a.ev) (void)hipEventRecord(a.ev[2], a.stream);
  return (int)hipGetLastError();
}

// The device records keep three bound values and nine cone entries per foot (mpc_core.h); the accessors hand out every bound and the
// dense cone block, formed here exactly as the solve kernel forms them (E times the bound).
template <int H>
void expand_records(const double *qp, const double *sc, double *xqp, double *xsc) {
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
      for (int k = 0; k < 9; ++k) xsc[C::XSC_AS + 15 * f + kAsPos[k]] = C::scaled_cone_entry(qp, sc, f, k);
      for (int r = 0; r < 5; ++r) {
        const double e = sc[C::SC_E + 5 * f + r];
        xsc[C::XSC_LS + 5 * f + r] = e * (r < 4 ? 0.0 : qp[C::QP_BND + 3 * f]);
        xsc[C::XSC_US + 5 * f + r] = e * (r < 4 ? qp[C::QP_BND + 3 * f + 1] : qp[C::QP_BND + 3 * f + 2]);
      }
    }
    for (int i = 0; i < 4; ++i) xsc[C::XSC_C + i] = sc[C::SC_C + i];
  }
}

}  // namespace

#define MPC_OPS_NAME2(h) mpc_horizon_ops_##h
#define MPC_OPS_NAME(h) MPC_OPS_NAME2(h)
extern "C" const mpc::HorizonOps *MPC_OPS_NAME(MPC_H)(void) {
  static const mpc::HorizonOps ops = {MPC_H, Cfg<MPC_H>::QP_LEN, Cfg<MPC_H>::SC_LEN, Cfg<MPC_H>::XQP_LEN, Cfg<MPC_H>::XSC_LEN, &launch<MPC_H>, &expand_records<MPC_H>};
  return &ops;
}
