// mpc_batch.hip -- the C ABI of include/mpc_batch.h: handles, record buffers, launches; the controller / FSM / policy kernels.
// The solver kernels are instantiated per planning horizon in mpc_horizon.hip (one translation unit per horizon, mpc_horizon.h) and
// reached through their HorizonOps entries.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/mpc_batch.h"
#include "controller.h"
#include "mpc_core.h"
#include "mpc_horizon.h"
#include "mpc_model.h"
#include "policy_mlp.h"

using namespace mpc;

#define MPC_DECL_OPS(HH) extern "C" const mpc::HorizonOps *mpc_horizon_ops_##HH(void);
MPC_HORIZON_LIST(MPC_DECL_OPS)
#undef MPC_DECL_OPS

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

constexpr int kWaitVm0 = 0x0F70;   // s_waitcnt vmcnt(0) (gfx9 encoding: vmcnt = simm16[3:0] | [15:14], expcnt [6:4] = 7, lgkmcnt [11:8] = 15: not waited for)

// the kernel set of a planning horizon, or null when the library was built without it (-DMPC_HORIZON_LIST)
const HorizonOps *horizon_ops(int h) {
#define MPC_ITEM(HH) if (h == HH) return mpc_horizon_ops_##HH();
  MPC_HORIZON_LIST(MPC_ITEM)
#undef MPC_ITEM
  return nullptr;
}

__global__ void reset_kernel(double *state, int state_len, const int *ids, int k, int n, int *seed, int seed_len) {
  const int r = blockIdx.x;
  const int robot = ids ? ids[r] : r;
  if (r >= k || robot < 0 || robot >= n) return;
  for (int i = threadIdx.x; i < state_len; i += blockDim.x) state[(size_t)robot * state_len + i] = 0.0;
  for (int i = threadIdx.x; i < seed_len; i += blockDim.x) seed[(size_t)robot * seed_len + i] = 0;      // (a new ConvexMpc object knows no working set either)
}

}  // namespace


constexpr int kTimingRing = 64;

struct mpc_batch {
  int n = 0, h = 0;
  const HorizonOps *ops = nullptr;   // the kernel set of this planning horizon (mpc_horizon.h)
  int state_len = 0;
  RobotModel *d_models = nullptr;
  double *d_state = nullptr, *d_qp = nullptr, *d_sc = nullptr;   // warm start, QP record (q, l, u, cone, wrench form of P), scale record
  int *d_info = nullptr;   // used when the caller passes no info buffer
  long long *d_prof = nullptr;   // per-robot section cycle counts of the last solve
  int *d_order = nullptr;        // job list of the solve kernel: the launch's active robots, longest expected solve first (order_block, written by the prep launch)
  int *d_sched = nullptr;        // [kSchedLen] job counters of the launch; d_ready [n]: polish entries (mpc_solve_jobs_kernel)
  int *d_ready = nullptr;
  int *d_seed = nullptr;         // [n, 4 h] exact mode: the working set each robot's previous call ended on (seeds the active-set method; MPC_EXACT_WARM=0: never)
  bool warm_sets = true;
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
// one solver launch on b's robots (+ the dispatch order for the next one); the input records as float32, float64 or float16
static int launch_solver(mpc_batch *b, const float *d_in, double *d_forces, int *d_info, const int *d_active, hipStream_t st, const double *d_in64 = nullptr,
                         const _Float16 *d_in16 = nullptr) {
  DeviceGuard guard_(b->device);
  const int slot = (int)(b->order_launches++ % kOrderHistory);
  if (b->exact) HIP_TRY(hipMemsetAsync(b->d_state, 0, sizeof(double) * (size_t)b->n * b->state_len, st));   // no warm start in that branch (mpc_osqp.cc:906-919)
  hipEvent_t *ev = b->timing ? b->ev[b->launches % kTimingRing] : nullptr;
  const LaunchArgs a{b->n, b->d_models, d_in, d_in64, d_in16, b->d_state, b->d_qp, b->d_sc, d_forces, d_info, b->d_prof, d_active, b->d_order, b->d_hist, slot, ev, st, b->exact,
                     b->max_iter, b->d_sched, b->d_ready, b->job_slots, b->warm_sets ? b->d_seed : nullptr};
  const hipError_t e = (hipError_t)b->ops->launch(a);
  if (e != hipSuccess) return fail(MPC_E_HIP, std::string("solver launch: ") + hipGetErrorString(e));
  b->launches++;
  return MPC_OK;
}

static int fetch_records(mpc_batch *b, double *h_qp, double *h_sc) {
  const size_t ql = b->ops->qp_len, sl = b->ops->sc_len, xql = b->ops->xqp_len, xsl = b->ops->xsc_len;
  std::vector<double> qp((size_t)b->n * ql), sc((size_t)b->n * sl);
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpy(qp.data(), b->d_qp, sizeof(double) * qp.size(), hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(sc.data(), b->d_sc, sizeof(double) * sc.size(), hipMemcpyDeviceToHost));
  for (int r = 0; r < b->n; ++r)
    b->ops->expand(qp.data() + (size_t)r * ql, sc.data() + (size_t)r * sl, h_qp ? h_qp + (size_t)r * xql : nullptr, h_sc ? h_sc + (size_t)r * xsl : nullptr);
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
  const HorizonOps *ops = horizon_ops(horizon);
  if (!ops) return fail(MPC_E_HORIZON, "mpc_batch_create: planning horizon outside the built range (mpc_supported_horizons: 2 .. 20 in the shipped library)");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail(MPC_E_NODEVICE, "mpc_batch_create: no HIP device");
  mpc_batch *b = new mpc_batch();
  (void)hipGetDevice(&b->device);
  b->n = n;
  b->h = horizon;
  b->ops = ops;
  const size_t qp_len = ops->qp_len, sc_len = ops->sc_len;
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
      (e = hipMalloc(&b->d_seed, sizeof(int) * (size_t)n * 4 * horizon)) != hipSuccess ||
      (e = hipMemset(b->d_seed, 0, sizeof(int) * (size_t)n * 4 * horizon)) != hipSuccess ||
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
    if (env) {   // (anything that is not a number, or negative, is ignored; fewer slots than one multi-wave workgroup needs means "not persistent")
      char *end = nullptr;
      const long v = strtol(env, &end, 10);
      if (end != env && *end == '\0' && v >= 0 && v <= (1 << 20)) b->job_slots = v < 2 ? 0 : (int)v;
    }
  }
  if (const char *ew = getenv("MPC_EXACT_WARM")) b->warm_sets = !(ew[0] == '0' && ew[1] == '\0');      // tuning hook: 0 = the exact mode's active-set method starts empty on every call
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
  if (b->d_seed) (void)hipFree(b->d_seed);
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
  const bool exact = solver == MPC_SOLVER_EXACT;
  if (exact != b->exact && b->d_seed) {      // a working set left by an earlier stretch in the exact mode is not this stretch's
    DeviceGuard guard_(b->device);
    HIP_TRY(hipDeviceSynchronize());      // (no stream argument here: solves in flight on any stream, non-blocking ones included, are over before the working sets go ...
    HIP_TRY(hipMemset(b->d_seed, 0, sizeof(int) * (size_t)b->n * 4 * b->h));
    HIP_TRY(hipDeviceSynchronize());      // ... and the clear is complete before a solve launched right after on such a stream can read them)
  }
  b->exact = exact;
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

int mpc_batch_solve_f16(mpc_batch *b, const unsigned short *d_in, double *d_forces, int *d_info, void *stream) {
  if (!b || !d_in || !d_forces) return fail(MPC_E_ARG, "mpc_batch_solve_f16: bad argument");
  return launch_solver(b, nullptr, d_forces, d_info ? d_info : b->d_info, nullptr, reinterpret_cast<hipStream_t>(stream), nullptr, reinterpret_cast<const _Float16 *>(d_in));
}

int mpc_batch_reset(mpc_batch *b, const int *ids, int k, void *stream) {
  if (!b) return fail(MPC_E_ARG, "mpc_batch_reset: bad argument");
  DeviceGuard guard_(b->device);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (!ids) {
    HIP_TRY(hipMemsetAsync(b->d_state, 0, sizeof(double) * (size_t)b->n * b->state_len, st));
    HIP_TRY(hipMemsetAsync(b->d_seed, 0, sizeof(int) * (size_t)b->n * 4 * b->h, st));
    return MPC_OK;
  }
  if (k <= 0) return MPC_OK;
  int *d_ids = nullptr;
  HIP_TRY(hipMallocAsync(reinterpret_cast<void **>(&d_ids), sizeof(int) * k, st));
  HIP_TRY(hipMemcpyAsync(d_ids, ids, sizeof(int) * k, hipMemcpyHostToDevice, st));
  hipLaunchKernelGGL(reset_kernel, dim3(k), dim3(256), 0, st, b->d_state, b->state_len, d_ids, k, b->n, b->d_seed, 4 * b->h);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipFreeAsync(d_ids, st));
  return MPC_OK;
}

int mpc_batch_reset_device(mpc_batch *b, const int *d_ids, int k, void *stream) {
  if (!b || !d_ids || k < 0) return fail(MPC_E_ARG, "mpc_batch_reset_device: bad argument");
  DeviceGuard guard_(b->device);
  if (k == 0) return MPC_OK;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(reset_kernel, dim3(k), dim3(256), 0, st, b->d_state, b->state_len, d_ids, k, b->n, b->d_seed, 4 * b->h);
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
int mpc_batch_qp_len(const mpc_batch *b) { return b ? b->ops->xqp_len : 0; }
int mpc_batch_scale_len(const mpc_batch *b) { return b ? b->ops->xsc_len : 0; }
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
  HIP_TRY(hipDeviceSynchronize());      // (the null stream does not order against non-blocking streams: nothing may be in flight on the state ...
  HIP_TRY(hipMemcpy(b->d_state, h_state, sizeof(double) * (size_t)b->n * b->state_len, hipMemcpyHostToDevice));
  HIP_TRY(hipMemset(b->d_seed, 0, sizeof(int) * (size_t)b->n * 4 * b->h));      // the working sets belonged to the state that was replaced
  HIP_TRY(hipDeviceSynchronize());      // ... and both are complete before the caller's next launch on any stream)
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
  if (r < n && gait_id[r] >= 0 && gait_id[r] < kNumGaitIds) st[r].gait_id = gait_id[r];      // (an id outside the dispatch leaves the robot's gait as it is)
}
__global__ void ctrl_set_iter_kernel(int n, CtrlState *st, const int *iter) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r < n) st[r].iter = iter[r];
}
// mpc_device_clock: one dependent v_fma_f64 chain per lane, one wave per SIMD on every CU (the solve kernel's own regime);
// shader cycles (s_memtime) of workgroup 0 over the HIP-event time of the launch = the shader clock the device grants under that load.
__global__ __launch_bounds__(64) void clock_probe_kernel(long long *out, int iters) {
  double a = threadIdx.x * 1e-3 + 1.0, b = 1.0000001, c = 1e-9;
  const long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int k = 0; k < 32; ++k) a = a * b + c;
  }
  const long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0 + (a == 0.5 ? 1 : 0);
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
// The four lanes of a quad each copy the whole st[r] and then store disjoint parts of it back.  Every lane's loads of st[r] must be
// complete before any lane of the quad stores: the quad sits in one wavefront (whose memory instructions issue in program order), so
// what is needed is that the COMPILER keeps every load above this point and every store below it, and that the loads' data has arrived
// (a lane may read a sibling's field late otherwise): wavefront-scope fence + s_waitcnt vmcnt(0) + a scheduling barrier.
static_assert(kCtrlThreads % 4 == 0, "a robot's four leg lanes must not straddle wavefronts");
__device__ __forceinline__ void quad_reads_done() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_s_waitcnt(kWaitVm0);
  __builtin_amdgcn_wave_barrier();
}
__global__ __launch_bounds__(kCtrlThreads) void ctrl_pre_kernel(int n, CtrlState *st, const RobotConst *rc, GaitTable gt, CtrlParams cp, const float *dof,
                                const float *est, const float *cmd, float *rec, int *active) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x, r = t >> 2, leg = t & 3;
  if (r >= n) return;           // (n robots = 4 n threads: whole quads leave together)
  CtrlState s = st[r];
  const RobotConst &k = rc[s.robot_type];
  ctrl_pre_legs(s, k, dof + (size_t)r * 24, leg, leg + 1);
  // every lane needs the foot positions of all four legs (centre-of-mass height, ground-normal fit, the solver record)
  const int q0 = (int)__lane_id() & ~3;      // first lane of my quad (four consecutive lanes of one wavefront: kCtrlThreads % 4 == 0)
  float fp[12];
#pragma unroll
  for (int l = 0; l < 4; ++l)
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float mine = c == 0 ? s.foot_positions[3 * leg] : (c == 1 ? s.foot_positions[3 * leg + 1] : s.foot_positions[3 * leg + 2]);
      fp[3 * l + c] = __shfl(mine, q0 + l);
    }
#pragma unroll
  for (int c = 0; c < 12; ++c) s.foot_positions[c] = fp[c];
  ctrl_pre_rest(s, k, gt, cp, est + (size_t)r * kEstLen, cmd + (size_t)r * 16, rec + (size_t)r * (56 + 4 * cp.horizon), leg, leg + 1, leg == 0);
  quad_reads_done();
  store_leg_fields(st[r], s, leg);
  if (leg == 0) {
    store_robot_fields(st[r], s);
    active[r] = s.do_solve;
  }
}
// controller.run's first half as ONE kernel of three concurrent wavefronts per 16 robots (mpc_ctrl_run; mpc_ctrl_step, whose caller brings the
// estimator outputs, keeps ctrl_pre_kernel):
//   wave 0   the leg quads of ctrl_pre_kernel: leg kinematics, then -- after the barrier -- everything of ctrl_pre_rest but the ground-normal fit;
//   wave 1   StateEstimator.update of the 16 robots, one lane each (estimator_kernel's work: no separate launch), handed over through LDS;
//   wave 2   the contact history and the ground-normal fit (gelsd43.h: ~20 us of one dependent float32 chain per robot) from the foot positions
//            wave 0 publishes, next to wave 0's foot placement / gait tables / record -- it writes the history, the normal and the record's normal.
// Every value is computed by the same code as in the two-kernel form (bit-identical results); what changes is that three dependent chains run side by side.
constexpr int kFusedRobots = 16;
__device__ __forceinline__ void store_leg_fields_no_hist(CtrlState &d, const CtrlState &s, int leg) {
  d.first_swing[leg] = s.first_swing[leg];
  d.swing_time_remaining[leg] = s.swing_time_remaining[leg];
  d.swing_times[leg] = s.swing_times[leg];
  d.contact_phase[leg] = s.contact_phase[leg];
  d.contact_states[leg] = s.contact_states[leg];
  d.swing_states[leg] = s.swing_states[leg];
  for (int c = 3 * leg; c < 3 * leg + 3; ++c) {
    d.f_ff[c] = s.f_ff[c]; d.p0[c] = s.p0[c]; d.pf[c] = s.pf[c]; d.tp[c] = s.tp[c]; d.tv[c] = s.tv[c];
    d.q[c] = s.q[c]; d.qd[c] = s.qd[c]; d.p[c] = s.p[c]; d.v[c] = s.v[c]; d.foot_positions[c] = s.foot_positions[c]; d.pfoot[c] = s.pfoot[c];
  }
  for (int c = 9 * leg; c < 9 * leg + 9; ++c) d.J[c] = s.J[c];
}
// WITH_POST: a tick on which the host knows that no robot is due for its MPC update (mpc_ctrl::mirror_valid) has no solver launch between the two halves,
// so wave 0 goes straight on into ctrl_post (swing / stance commands, torque map) with the state it holds: one kernel per tick instead of two.
template <bool WITH_POST>
__global__ __launch_bounds__(3 * 64) void ctrl_pre_fused_kernel(int n, CtrlState *st, const RobotConst *rc, GaitTable gt, CtrlParams cp, const float *dof, const float *body,
                                                               float *est_out, const float *cmd, float *rec, int *active, float *torques) {
  __shared__ float sh_est[kFusedRobots][kEstLen];
  __shared__ float sh_fp[kFusedRobots][12];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, r0 = blockIdx.x * kFusedRobots;
  if (wave == 1) {
    const int r = r0 + lane;
    if (lane < kFusedRobots && r < n) {
      const float nrm[3] = {st[r].normal[0], st[r].normal[1], st[r].normal[2]};       // the estimate of the previous tick (StateEstimator.py:88-92)
      float e[kEstLen];
      estimator_update(body + (size_t)r * 13, nrm, e);
      for (int k = 0; k < kEstLen; ++k) { sh_est[lane][k] = e[k]; est_out[(size_t)r * kEstLen + k] = e[k]; }
    }
    __builtin_amdgcn_s_waitcnt(kWaitVm0);      // (like waves 0 and 2: this wave's loads of the state -- the previous tick's normal -- have arrived before wave 2 may overwrite it)
    __syncthreads();
    return;
  }
  if (wave == 2) {
    const int r = r0 + lane;
    const bool on = lane < kFusedRobots && r < n && !cp.flat_ground;
    float hist[12], cph[4], nrm[3];
    int first_run = 0, due = 0;
    double body_height = 0.0;
    if (on) {
      for (int k = 0; k < 12; ++k) hist[k] = st[r].hist[k];
      for (int k = 0; k < 4; ++k) cph[k] = st[r].contact_phase[k];
      first_run = st[r].first_run;
      due = ((st[r].iter + 1) % cp.iters_between_mpc) == 0;      // ctrl_pre_rest's do_solve of this tick: iterationCounter += 1, then the test (wave 0 stores the counter after the barrier)
      body_height = rc[st[r].robot_type].body_height;
    }
    __builtin_amdgcn_s_waitcnt(kWaitVm0);      // (the loads of the state are complete before wave 0 may store to it: it stores after the barrier)
    __syncthreads();
    if (on) {
      float fp[12];
      for (int k = 0; k < 12; ++k) fp[k] = sh_fp[lane][k];
      if (first_run) contact_history_init(hist, fp, body_height);
      ground_normal_update(hist, nrm, cph, fp);
      for (int k = 0; k < 12; ++k) st[r].hist[k] = hist[k];
      for (int k = 0; k < 3; ++k) st[r].normal[k] = nrm[k];
      if (due)      // the solver record is written on MPC ticks only, like the two-kernel path's (controller.h ctrl_pre_rest)
        for (int k = 0; k < 3; ++k) rec[(size_t)r * (56 + 4 * cp.horizon) + IN_NRM + k] = nrm[k];
    }
    return;
  }
  const int rl = lane >> 2, leg = lane & 3, r = r0 + rl;
  const bool on = r < n;
  CtrlState s;
  if (on) {
    s = st[r];
    ctrl_pre_legs(s, rc[s.robot_type], dof + (size_t)r * 24, leg, leg + 1);
    for (int c = 0; c < 3; ++c) sh_fp[rl][3 * leg + c] = c == 0 ? s.foot_positions[3 * leg] : (c == 1 ? s.foot_positions[3 * leg + 1] : s.foot_positions[3 * leg + 2]);
  }
  __builtin_amdgcn_s_waitcnt(kWaitVm0);
  __syncthreads();
  if (!on) return;
  const RobotConst &k = rc[s.robot_type];
  float e[kEstLen];
  for (int c = 0; c < 12; ++c) s.foot_positions[c] = sh_fp[rl][c];
  for (int c = 0; c < kEstLen; ++c) e[c] = sh_est[rl][c];
  ctrl_pre_rest(s, k, gt, cp, e, cmd + (size_t)r * 16, rec + (size_t)r * (56 + 4 * cp.horizon), leg, leg + 1, leg == 0, false);
  if constexpr (WITH_POST) ctrl_post(s, k, nullptr, 0, torques + (size_t)r * 12, leg, leg + 1);      // (do_solve is 0 for every robot: no forces are read)
  quad_reads_done();
  if (cp.flat_ground) store_leg_fields(st[r], s, leg);      // (no fit on flat ground: the history is this wave's, set once at the first run)
  else store_leg_fields_no_hist(st[r], s, leg);
  if (leg == 0) {
    st[r].iter = s.iter; st[r].first_run = s.first_run; st[r].pos_z = s.pos_z; st[r].posz_tick = s.posz_tick; st[r].do_solve = s.do_solve;
    for (int c = 0; c < 3; ++c) st[r].vbody[c] = s.vbody[c];
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
                                                                double *solver_state, int state_len, int *seed, int seed_len) {
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
  if (f.entered_loco && solver_state) {
    for (int q = 0; q < state_len; ++q) solver_state[(size_t)r * state_len + q] = 0.0;
    for (int q = 0; q < seed_len; ++q) seed[(size_t)r * seed_len + q] = 0;           // ... which knows no working set either (exact mode)
  }
  st[r] = s; fs[r] = f;
}
// RobotRunnerFSM.run up to the solver launch: fsm_tick, then ctrl_pre for the robots whose state runs the locomotion controller
__global__ __launch_bounds__(kCtrlThreads) void fsm_pre_kernel(int n, CtrlState *st, FsmState *fs, const RobotConst *rc, GaitTable gt, CtrlParams cp, FsmParams fp,
                               const float *dof, const float *body, const float *est, const float *cmd, const int *request, float *rec,
                               int *active, double *solver_state, int state_len, int *seed, int seed_len) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n) return;
  CtrlState s = st[r];
  FsmState f = fs[r];
  fsm_tick(f, s, rc[s.robot_type], fp, dof + (size_t)r * 24, body + (size_t)r * 13, request[r]);
  if (f.entered_loco) {  // cMPC.initialize built a new ConvexMpc (ConvexMPCLocomotion.py:102-108): the next solve is a cold one, from the empty working set
    for (int k = 0; k < state_len; ++k) solver_state[(size_t)r * state_len + k] = 0.0;
    for (int k = 0; k < seed_len; ++k) seed[(size_t)r * seed_len + k] = 0;
  }
  int act = 0;
  if (f.run_loco) {
    ctrl_pre(s, rc[s.robot_type], gt, cp, dof + (size_t)r * 24, est + (size_t)r * kEstLen, cmd + (size_t)r * 16, rec + (size_t)r * (56 + 4 * cp.horizon));
    act = s.do_solve;
  }
  active[r] = act;
  st[r] = s; fs[r] = f;
}
// Does the controller take the solver's forces?  OSQP branch: only OSQP_SOLVED returns a vector (mpc_osqp.cc:781-794; with the empty list the
// reference's Python raises at ConvexMPCLocomotion.py:186-187 -- here f_ff keeps its value).  qpOASES branch (exact mode): the vector comes
// back whatever the solver's status (:906-947) and ConvexMPCLocomotion.py:186-187 adopts it unconditionally: every status for which the
// library wrote the row (SOLVED, SOLVED_INACCURATE, MAX_ITER_REACHED: mpc_wrench.h store()) -- only NON_CVX writes nothing.
__device__ __forceinline__ int adopts_forces(int status, int exact) {
  return exact ? (status == kStSolved || status == kStSolvedInaccurate || status == kStMaxIter) : status == kStSolved;
}
__global__ __launch_bounds__(kCtrlThreads) void fsm_post_kernel(int n, CtrlState *st, const FsmState *fs, const RobotConst *rc, int horizon, const double *forces, const int *info,
                                int exact, float *torques) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n) return;
  if (fs[r].run_loco) {
    CtrlState s = st[r];
    ctrl_post(s, rc[s.robot_type], forces + (size_t)r * 12 * horizon, adopts_forces(info[(size_t)r * kInfoLen + 1], exact), torques + (size_t)r * 12);
    st[r] = s;
  } else {
    fsm_joint_torques(fs[r], st[r], torques + (size_t)r * 12);
  }
}
__global__ __launch_bounds__(kCtrlThreads) void ctrl_post_kernel(int n, CtrlState *st, const RobotConst *rc, int horizon, const double *forces, const int *info,
                                 int exact, float *torques) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x, r = t >> 2, leg = t & 3;      // one lane per leg (see ctrl_pre_kernel)
  if (r >= n) return;
  CtrlState s = st[r];
  ctrl_post(s, rc[s.robot_type], forces + (size_t)r * 12 * horizon, adopts_forces(info[(size_t)r * kInfoLen + 1], exact), torques + (size_t)r * 12, leg, leg + 1);
  quad_reads_done();
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

// one tick: the pre kernel (with the caller's estimator outputs, or -- d_est null -- with StateEstimator.update from d_body fused in), the solver
// launch when a robot is due, the post kernel
static int ctrl_tick(mpc_ctrl *c, const float *d_dof, const float *d_est, const float *d_body, const float *d_cmd, float *d_torques, hipStream_t st) {
  const int n = c->n, blocks4 = (4 * n + kCtrlThreads - 1) / kCtrlThreads;      // (ctrl_pre / ctrl_post: one lane per leg)
  bool any_due = true;
  if (c->mirror_valid) {   // ConvexMPCLocomotion.run: iterationCounter += 1, MPC update when it is a multiple of iterationsBetweenMPC
    if ((int)c->h_iter.size() != n) c->h_iter.assign(n, 0);
    any_due = false;
    for (int r = 0; r < n; ++r) any_due |= (++c->h_iter[r] % c->cp.iters_between_mpc) == 0;
  }
  if (d_est) hipLaunchKernelGGL(ctrl_pre_kernel, dim3(blocks4), dim3(kCtrlThreads), 0, st, n, c->d_state, c->d_rc, c->gt, c->cp, d_dof, d_est, d_cmd, c->d_rec, c->d_active);
  else if (!any_due) {      // nobody is due (host mirror): the whole tick is one kernel
    hipLaunchKernelGGL(ctrl_pre_fused_kernel<true>, dim3((n + kFusedRobots - 1) / kFusedRobots), dim3(3 * 64), 0, st, n, c->d_state, c->d_rc, c->gt, c->cp, d_dof, d_body, c->d_est,
                       d_cmd, c->d_rec, c->d_active, d_torques);
    HIP_TRY(hipGetLastError());
    return MPC_OK;
  }
  else hipLaunchKernelGGL(ctrl_pre_fused_kernel<false>, dim3((n + kFusedRobots - 1) / kFusedRobots), dim3(3 * 64), 0, st, n, c->d_state, c->d_rc, c->gt, c->cp, d_dof, d_body, c->d_est,
                          d_cmd, c->d_rec, c->d_active, (float *)nullptr);
  HIP_TRY(hipGetLastError());
  mpc_batch *b = c->solver;
  if (any_due) {
    int rc = launch_solver(b, c->d_rec, c->d_forces, c->d_info, c->d_active, st);
    if (rc != MPC_OK) return rc;
  }
  hipLaunchKernelGGL(ctrl_post_kernel, dim3(blocks4), dim3(kCtrlThreads), 0, st, n, c->d_state, c->d_rc, c->cp.horizon, c->d_forces, c->d_info, b->exact, d_torques);
  HIP_TRY(hipGetLastError());
  return MPC_OK;
}

int mpc_ctrl_step(mpc_ctrl *c, const float *d_dof, const float *d_est, const float *d_cmd, float *d_torques, void *stream) {
  if (!c || !d_dof || !d_est || !d_cmd || !d_torques) return fail(MPC_E_ARG, "mpc_ctrl_step: bad argument");
  DeviceGuard guard_(c->solver->device);
  return ctrl_tick(c, d_dof, d_est, nullptr, d_cmd, d_torques, reinterpret_cast<hipStream_t>(stream));
}

int mpc_ctrl_run(mpc_ctrl *c, const float *d_dof, const float *d_body, const float *d_cmd, float *d_torques, void *stream) {
  if (!c || !d_dof || !d_body || !d_cmd || !d_torques) return fail(MPC_E_ARG, "mpc_ctrl_run: bad argument");
  DeviceGuard guard_(c->solver->device);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  return ctrl_tick(c, d_dof, nullptr, d_body, d_cmd, d_torques, st);
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

int mpc_ctrl_set_gait_device(mpc_ctrl *c, const int *d_gait_id, void *stream) {
  if (!c || !d_gait_id) return fail(MPC_E_ARG, "mpc_ctrl_set_gait_device: bad argument");
  DeviceGuard guard_(c->solver->device);
  hipLaunchKernelGGL(ctrl_set_gait_kernel, dim3((c->n + kCtrlThreads - 1) / kCtrlThreads), dim3(kCtrlThreads), 0, reinterpret_cast<hipStream_t>(stream), c->n, c->d_state, d_gait_id);
  HIP_TRY(hipGetLastError());
  return MPC_OK;      // (stream-ordered: no host round trip, no synchronisation)
}

int mpc_ctrl_set_iteration(mpc_ctrl *c, const int *iteration, void *stream) {
  if (!c || !iteration) return fail(MPC_E_ARG, "mpc_ctrl_set_iteration: bad argument");
  DeviceGuard guard_(c->solver->device);
  for (int r = 0; r < c->n; ++r) if (iteration[r] < 0) return fail(MPC_E_ARG, "mpc_ctrl_set_iteration: negative counter");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  HIP_TRY(hipMemcpyAsync(c->d_active, iteration, sizeof(int) * c->n, hipMemcpyHostToDevice, st));   // (d_active is rewritten by the next tick's ctrl_pre)
  hipLaunchKernelGGL(ctrl_set_iter_kernel, dim3((c->n + kCtrlThreads - 1) / kCtrlThreads), dim3(kCtrlThreads), 0, st, c->n, c->d_state, c->d_active);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(st));      // `iteration` is a host buffer
  c->h_iter.assign(iteration, iteration + c->n);
  return MPC_OK;
}

mpc_batch *mpc_ctrl_solver(mpc_ctrl *c) { return c ? c->solver : nullptr; }

int mpc_ctrl_solver_forces(mpc_ctrl *c, double *h_forces) {
  if (!c || !h_forces) return fail(MPC_E_ARG, "mpc_ctrl_solver_forces: bad argument");
  DeviceGuard guard_(c->solver->device);
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpy(h_forces, c->d_forces, sizeof(double) * (size_t)c->n * 12 * c->cp.horizon, hipMemcpyDeviceToHost));
  return MPC_OK;
}

int mpc_device_clock(int device, int busy_ms, double *ghz, double *ms) {
  if (!ghz || busy_ms <= 0 || busy_ms > 1000) return fail(MPC_E_ARG, "mpc_device_clock: bad argument (1 .. 1000 ms)");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return fail(MPC_E_NODEVICE, "mpc_device_clock: no such HIP device");
  DeviceGuard guard_(device);
  hipDeviceProp_t prop;
  HIP_TRY(hipGetDeviceProperties(&prop, device));
  const int blocks = 4 * prop.multiProcessorCount;       // one single-wave workgroup per SIMD
  struct Probe {      // released on every path out
    long long *d_out = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    hipStream_t st = nullptr;
    ~Probe() {
      if (e0) (void)hipEventDestroy(e0);
      if (e1) (void)hipEventDestroy(e1);
      if (d_out) (void)hipFree(d_out);
      if (st) (void)hipStreamDestroy(st);
    }
  } pr;
  HIP_TRY(hipMalloc(&pr.d_out, sizeof(long long) * blocks));
  HIP_TRY(hipStreamCreateWithFlags(&pr.st, hipStreamNonBlocking));      // a stream of its own: the null stream would serialise with the caller's work
  HIP_TRY(hipEventCreate(&pr.e0)); HIP_TRY(hipEventCreate(&pr.e1));
  // ~6.3 shader cycles per dependent fp64 FMA, 32 per trip: ~10 k trips per ms at 2.1 GHz
  hipLaunchKernelGGL(clock_probe_kernel, dim3(blocks), dim3(64), 0, pr.st, pr.d_out, 2000);
  HIP_TRY(hipEventRecord(pr.e0, pr.st));
  hipLaunchKernelGGL(clock_probe_kernel, dim3(blocks), dim3(64), 0, pr.st, pr.d_out, 10000 * busy_ms);
  HIP_TRY(hipEventRecord(pr.e1, pr.st));
  HIP_TRY(hipEventSynchronize(pr.e1));
  float t = 0.f;
  HIP_TRY(hipEventElapsedTime(&t, pr.e0, pr.e1));
  long long cyc = 0;
  HIP_TRY(hipMemcpyAsync(&cyc, pr.d_out, sizeof cyc, hipMemcpyDeviceToHost, pr.st));
  HIP_TRY(hipStreamSynchronize(pr.st));
  *ghz = t > 0.f ? (double)cyc / ((double)t * 1e6) : 0.0;
  if (ms) *ms = t;
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
                     (const int *)nullptr, c->n, 1, (double *)nullptr, 0, (int *)nullptr, 0);   // (the whole solver was reset above)
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
                     c->solver->d_state, c->solver->state_len, c->solver->d_seed, 4 * c->solver->h);
  HIP_TRY(hipGetLastError());
  if (d_ids) HIP_TRY(hipFreeAsync(d_ids, st));
  HIP_TRY(hipStreamSynchronize(st));
  return MPC_OK;
}

int mpc_ctrl_fsm_reset_device(mpc_ctrl *c, const int *d_ids, int k, void *stream) {
  if (!c || !c->d_fsm || !d_ids || k < 0) return fail(MPC_E_ARG, "mpc_ctrl_fsm_reset_device: bad argument (mpc_ctrl_fsm_init first)");
  DeviceGuard guard_(c->solver->device);
  if (k == 0) return MPC_OK;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(fsm_init_kernel, dim3((k + kCtrlThreads - 1) / kCtrlThreads), dim3(kCtrlThreads), 0, st, c->n, c->d_state, c->d_fsm, c->d_rc, c->d_fsm_mode, c->fsm_op_mode, d_ids, k, 0,
                     c->solver->d_state, c->solver->state_len, c->solver->d_seed, 4 * c->solver->h);
  HIP_TRY(hipGetLastError());
  return MPC_OK;      // (stream-ordered: no host round trip, no synchronisation)
}

static int run_fsm(mpc_ctrl *c, const float *d_dof, const float *d_body, const float *d_cmd, const int *d_request, float *d_torques, void *stream, bool estimated) {
  if (!c || !d_dof || !d_body || !d_cmd || !d_request || !d_torques) return fail(MPC_E_ARG, "mpc_ctrl_run_fsm: bad argument");
  DeviceGuard guard_(c->solver->device);
  if (!c->d_fsm) return fail(MPC_E_ARG, "mpc_ctrl_run_fsm: call mpc_ctrl_fsm_init first");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int n = c->n, blocks = (n + kCtrlThreads - 1) / kCtrlThreads;
  mpc_batch *b = c->solver;
  if (!estimated) hipLaunchKernelGGL(estimator_kernel, dim3(blocks), dim3(kCtrlThreads), 0, st, n, c->d_state, d_body, c->d_est);
  hipLaunchKernelGGL(fsm_pre_kernel, dim3(blocks), dim3(kCtrlThreads), 0, st, n, c->d_state, c->d_fsm, c->d_rc, c->gt, c->cp, c->fp, d_dof, d_body, c->d_est, d_cmd,
                     d_request, c->d_rec, c->d_active, b->d_state, b->state_len, b->d_seed, 4 * b->h);
  HIP_TRY(hipGetLastError());
  int rc = launch_solver(b, c->d_rec, c->d_forces, c->d_info, c->d_active, st);
  if (rc != MPC_OK) return rc;
  hipLaunchKernelGGL(fsm_post_kernel, dim3(blocks), dim3(kCtrlThreads), 0, st, n, c->d_state, c->d_fsm, c->d_rc, c->cp.horizon, c->d_forces, c->d_info, b->exact, d_torques);
  HIP_TRY(hipGetLastError());
  return MPC_OK;
}
int mpc_ctrl_run_fsm(mpc_ctrl *c, const float *d_dof, const float *d_body, const float *d_cmd, const int *d_request, float *d_torques, void *stream) {
  return run_fsm(c, d_dof, d_body, d_cmd, d_request, d_torques, stream, false);
}
int mpc_ctrl_run_fsm_estimated(mpc_ctrl *c, const float *d_dof, const float *d_body, const float *d_cmd, const int *d_request, float *d_torques, void *stream) {
  return run_fsm(c, d_dof, d_body, d_cmd, d_request, d_torques, stream, true);
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

int mpc_ctrl_solver_record(mpc_ctrl *c, float *h_rec) {
  if (!c || !h_rec) return fail(MPC_E_ARG, "mpc_ctrl_solver_record: bad argument");
  DeviceGuard guard_(c->solver->device);
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpy(h_rec, c->d_rec, sizeof(float) * (size_t)c->n * (56 + 4 * (size_t)c->cp.horizon), hipMemcpyDeviceToHost));
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
    if (dims[l] <= 0 || dims[l + 1] <= 0 || dims[l] % 16 != 0 || !weights[l] || !biases[l])
      return fail(MPC_E_ARG, "mpc_policy_create: layer input widths must be positive multiples of 16");
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
  hipLaunchKernelGGL(policy::observations_kernel, dim3((n * 48 + 255) / 256), dim3(256), 0, (hipStream_t)stream, n, d_dof, d_est, d_normal, 3,
                     d_cmd3, d_prev_actions, scales4[0], scales4[1], scales4[2], scales4[3], d_obs);
  HIP_TRY(hipGetLastError());
  return MPC_OK;
}

int mpc_ctrl_policy_observations(mpc_ctrl *c, const float *d_dof, const float *d_cmd3, const float *d_prev_actions, const float *scales4, float *d_obs, void *stream) {
  if (!c || !d_dof || !d_cmd3 || !d_prev_actions || !scales4 || !d_obs) return fail(MPC_E_ARG, "mpc_ctrl_policy_observations: bad argument");
  DeviceGuard guard_(c->solver->device);
  static_assert(sizeof(CtrlState) % sizeof(float) == 0, "the state records are read with a float stride");
  hipLaunchKernelGGL(policy::observations_kernel, dim3((c->n * 48 + 255) / 256), dim3(256), 0, (hipStream_t)stream, c->n, d_dof, c->d_est, &c->d_state[0].normal[0],
                     (int)(sizeof(CtrlState) / sizeof(float)), d_cmd3, d_prev_actions, scales4[0], scales4[1], scales4[2], scales4[3], d_obs);
  HIP_TRY(hipGetLastError());
  return MPC_OK;
}

int mpc_pack_commands_scaled(int n, const float *d_cmd3, const float *d_actions12, const float *scale12, const float *const12, float *d_cmd16, void *stream) {
  if (n <= 0 || !d_cmd3 || !d_actions12 || !scale12 || !const12 || !d_cmd16) return fail(MPC_E_ARG, "mpc_pack_commands_scaled: bad argument");
  policy::Rescale rs;
  for (int k = 0; k < 12; ++k) { rs.scale[k] = scale12[k]; rs.shift[k] = const12[k]; }
  hipLaunchKernelGGL(policy::pack_commands_scaled_kernel, dim3((n * 16 + 255) / 256), dim3(256), 0, (hipStream_t)stream, n, d_cmd3, d_actions12, rs, d_cmd16);
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
