// mpc_peer.hip -- the optional torque exchange of SURVEY.md 8(e) as ONE-SHOT DIRECT PEER WRITES (a 24 KB message per GPU is latency-bound; a ring
// collective pays one hop per rank): every rank holds a receive buffer for the whole env batch, exported to the other ranks of the node as a hipIpc handle; "put"
// is one kernel that stores this rank's rows straight into every rank's buffer over xGMI and then raises this rank's epoch flag there (system-scope release),
// "wait" one kernel that watches the local flags (system-scope acquire, bounded).  No collective library, no host synchronisation; the alternative to the RCCL
// all-gather behind sharding.ShardedLocomotion.start_gather / torques_all.  UNMEASURED ACROSS GPUs (no multi-GPU node was available to any round): the tests run it
// between two processes of ONE GPU (the IPC path, the flags, the double buffering) and in a one-rank group.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/mpc_batch.h"

namespace {
thread_local std::string g_perr;
int pfail(int code, const std::string &m) { g_perr = m; return code; }
#define PEER_TRY(expr)                                                                               \
  do {                                                                                               \
    hipError_t e_ = (expr);                                                                          \
    if (e_ != hipSuccess) return pfail(MPC_E_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
  } while (0)

constexpr int kMaxRanks = 16;
typedef unsigned v4u __attribute__((ext_vector_type(4)));
struct Peers { unsigned char *buf[kMaxRanks]; };

// layout of a rank's region: [2 parities][n_total * row_bytes] data, then [2][kMaxRanks] uint32 epoch flags
__global__ void peer_put_kernel(Peers peers, int world, int rank, const unsigned char *__restrict__ local, size_t data_off, size_t nbytes, size_t flag_off, unsigned epoch) {
  // blockIdx.y = destination rank; 16-byte stores where the block allows
  const int dst = blockIdx.y;
  unsigned char *out = peers.buf[dst] + data_off;
  const size_t i0 = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 16, stride = (size_t)gridDim.x * blockDim.x * 16;
  for (size_t i = i0; i + 16 <= nbytes; i += stride) __builtin_nontemporal_store(*reinterpret_cast<const v4u *>(local + i), reinterpret_cast<v4u *>(out + i));
  if (blockIdx.x == 0 && threadIdx.x < (nbytes & 15)) out[(nbytes & ~(size_t)15) + threadIdx.x] = local[(nbytes & ~(size_t)15) + threadIdx.x];
  // the flag: after EVERY store of this destination (all of its workgroups): a per-destination arrival counter in my own region
  __threadfence_system();
  __shared__ int last;
  unsigned *arrive = reinterpret_cast<unsigned *>(peers.buf[rank] + flag_off) + 2 * kMaxRanks + dst;      // (scratch counters behind the flags)
  __syncthreads();
  if (threadIdx.x == 0) last = atomicAdd(arrive, 1u) == gridDim.x - 1;
  __syncthreads();
  if (last && threadIdx.x == 0) {
    *arrive = 0;
    unsigned *flag = reinterpret_cast<unsigned *>(peers.buf[dst] + flag_off) + (epoch & 1) * kMaxRanks + rank;
    __hip_atomic_store(flag, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// every workgroup waits for all ranks' flags of this epoch itself, then copies its share of the batch out of the receive region
__global__ void peer_wait_kernel(const unsigned char *mine, int world, size_t data_off, size_t nbytes, size_t flag_off, unsigned epoch, long long max_cycles, int *timeouts,
                                 unsigned char *__restrict__ out) {
  const int r = threadIdx.x;
  if (r < world) {
    const unsigned *flag = reinterpret_cast<const unsigned *>(mine + flag_off) + (epoch & 1) * kMaxRanks + r;
    const long long t0 = wall_clock64();
    // (epochs only grow; a later epoch of the same parity cannot arrive before this rank has waited for this one: see mpc_batch.h)
    while ((int)(__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) - epoch) < 0) {
      if (wall_clock64() - t0 > max_cycles) { if (blockIdx.x == 0) atomicAdd(timeouts, 1); break; }
      __builtin_amdgcn_s_sleep(8);
    }
  }
  __syncthreads();
  if (!out) return;
  const unsigned char *in = mine + data_off;
  const size_t i0 = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 16, stride = (size_t)gridDim.x * blockDim.x * 16;
  for (size_t i = i0; i + 16 <= nbytes; i += stride) *reinterpret_cast<v4u *>(out + i) = __builtin_nontemporal_load(reinterpret_cast<const v4u *>(in + i));
  if (blockIdx.x == 0 && threadIdx.x < (nbytes & 15)) out[(nbytes & ~(size_t)15) + threadIdx.x] = in[(nbytes & ~(size_t)15) + threadIdx.x];
}
}  // namespace

struct mpc_peer {
  int rank = 0, world = 1, device = 0;
  size_t rows = 0, row_bytes = 0, data_bytes = 0, flag_off = 0, total = 0;
  unsigned char *mine = nullptr;
  Peers peers{};
  bool opened[kMaxRanks] = {};
  unsigned epoch = 0;          // of the last put
  int *d_timeouts = nullptr;
  double timeout_s = 2.0;
};

extern "C" {

const char *mpc_peer_last_error(void) { return g_perr.c_str(); }

int mpc_peer_create(mpc_peer **out, int rank, int world, int n_rows_total, int row_bytes) {
  if (!out || world < 1 || world > kMaxRanks || rank < 0 || rank >= world || n_rows_total <= 0 || row_bytes <= 0) return pfail(MPC_E_ARG, "mpc_peer_create: bad argument (at most 16 ranks)");
  mpc_peer *p = new mpc_peer();
  p->rank = rank; p->world = world; p->rows = (size_t)n_rows_total; p->row_bytes = (size_t)row_bytes;
  p->data_bytes = ((p->rows * p->row_bytes + 255) / 256) * 256;
  p->flag_off = 2 * p->data_bytes;
  p->total = p->flag_off + sizeof(unsigned) * 3 * kMaxRanks;
  if (hipGetDevice(&p->device) != hipSuccess) { delete p; return pfail(MPC_E_NODEVICE, "mpc_peer_create: no HIP device"); }
  hipError_t e;
  // fine-grained device memory: peers' stores and this device's loads meet without a kernel boundary in between
  if ((e = hipExtMallocWithFlags(reinterpret_cast<void **>(&p->mine), p->total, hipDeviceMallocFinegrained)) != hipSuccess ||
      (e = hipMemset(p->mine, 0, p->total)) != hipSuccess || (e = hipMalloc(&p->d_timeouts, sizeof(int))) != hipSuccess ||
      (e = hipMemset(p->d_timeouts, 0, sizeof(int))) != hipSuccess || (e = hipDeviceSynchronize()) != hipSuccess) {
    if (p->mine) (void)hipFree(p->mine);
    if (p->d_timeouts) (void)hipFree(p->d_timeouts);
    delete p;
    return pfail(MPC_E_HIP, std::string("mpc_peer_create: ") + hipGetErrorString(e));
  }
  p->peers.buf[rank] = p->mine;
  *out = p;
  return MPC_OK;
}

int mpc_peer_handle(mpc_peer *p, void *handle64) {
  if (!p || !handle64) return pfail(MPC_E_ARG, "mpc_peer_handle: bad argument");
  static_assert(sizeof(hipIpcMemHandle_t) == MPC_PEER_HANDLE_BYTES, "the IPC handle is passed around as 64 bytes");
  hipIpcMemHandle_t h;
  PEER_TRY(hipIpcGetMemHandle(&h, p->mine));
  std::memcpy(handle64, &h, sizeof h);
  return MPC_OK;
}

int mpc_peer_connect(mpc_peer *p, const void *handles) {
  if (!p || (!handles && p->world > 1)) return pfail(MPC_E_ARG, "mpc_peer_connect: bad argument");
  for (int r = 0; r < p->world; ++r) {
    if (r == p->rank || p->opened[r]) continue;
    hipIpcMemHandle_t h;
    std::memcpy(&h, static_cast<const unsigned char *>(handles) + (size_t)r * sizeof h, sizeof h);
    void *ptr = nullptr;
    PEER_TRY(hipIpcOpenMemHandle(&ptr, h, hipIpcMemLazyEnablePeerAccess));
    p->peers.buf[r] = static_cast<unsigned char *>(ptr);
    p->opened[r] = true;
  }
  return MPC_OK;
}

int mpc_peer_put(mpc_peer *p, const void *d_local, int row_lo, int n_rows, void *stream) {
  if (!p || !d_local || row_lo < 0 || n_rows < 0 || (size_t)row_lo + (size_t)n_rows > p->rows) return pfail(MPC_E_ARG, "mpc_peer_put: bad argument");
  for (int r = 0; r < p->world; ++r) if (!p->peers.buf[r]) return pfail(MPC_E_ARG, "mpc_peer_put: mpc_peer_connect first");
  if ((reinterpret_cast<uintptr_t>(d_local) & 15) || ((size_t)row_lo * p->row_bytes & 15)) return pfail(MPC_E_ARG, "mpc_peer_put: the block must start on a 16-byte boundary");
  const unsigned epoch = ++p->epoch;
  const size_t nbytes = (size_t)n_rows * p->row_bytes, off = (epoch & 1) * p->data_bytes + (size_t)row_lo * p->row_bytes;
  int blocks = (int)((nbytes / 16 + 255) / 256);
  blocks = blocks < 1 ? 1 : (blocks > 64 ? 64 : blocks);
  hipLaunchKernelGGL(peer_put_kernel, dim3(blocks, p->world), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), p->peers, p->world, p->rank,
                     static_cast<const unsigned char *>(d_local), off, nbytes, p->flag_off, epoch);
  PEER_TRY(hipGetLastError());
  return MPC_OK;
}

int mpc_peer_wait(mpc_peer *p, void *d_out, void *stream) {
  if (!p) return pfail(MPC_E_ARG, "mpc_peer_wait: bad argument");
  if (p->epoch == 0) return pfail(MPC_E_ARG, "mpc_peer_wait: mpc_peer_put first");
  if (reinterpret_cast<uintptr_t>(d_out) & 15) return pfail(MPC_E_ARG, "mpc_peer_wait: the output must start on a 16-byte boundary");
  const long long max_cycles = (long long)(p->timeout_s * 100e6);      // wall_clock64: 100 MHz
  const size_t nbytes = p->rows * p->row_bytes;
  int blocks = (int)((nbytes / 16 + 255) / 256);
  blocks = !d_out || blocks < 1 ? 1 : (blocks > 64 ? 64 : blocks);
  hipLaunchKernelGGL(peer_wait_kernel, dim3(blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), p->mine, p->world, (p->epoch & 1) * p->data_bytes, nbytes, p->flag_off, p->epoch,
                     max_cycles, p->d_timeouts, static_cast<unsigned char *>(d_out));
  PEER_TRY(hipGetLastError());
  return MPC_OK;
}

int mpc_peer_timeouts(mpc_peer *p, int *count) {
  if (!p || !count) return pfail(MPC_E_ARG, "mpc_peer_timeouts: bad argument");
  PEER_TRY(hipDeviceSynchronize());
  PEER_TRY(hipMemcpy(count, p->d_timeouts, sizeof(int), hipMemcpyDeviceToHost));
  return MPC_OK;
}

void mpc_peer_destroy(mpc_peer *p) {
  if (!p) return;
  (void)hipDeviceSynchronize();
  for (int r = 0; r < p->world; ++r) if (p->opened[r]) (void)hipIpcCloseMemHandle(p->peers.buf[r]);
  if (p->mine) (void)hipFree(p->mine);
  if (p->d_timeouts) (void)hipFree(p->d_timeouts);
  delete p;
}

}  // extern "C"
