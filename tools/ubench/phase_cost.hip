// Microbenchmark: cost of one barrier-separated "phase" (LDS read -> NDEP dependent fp64 ops -> LDS write -> s_barrier)
// for a 256-thread workgroup, with one and with two workgroups resident per CU.
// build: hipcc --offload-arch=gfx950 -O3 phase_cost.hip -o phase_cost
#include <hip/hip_runtime.h>
#include <cstdio>
template <int NDEP, int NLOAD>
__global__ __launch_bounds__(256) void k(long long *out, double *sink, int phases, int lds_pad) {
  extern __shared__ double lds[];
  const int tid = threadIdx.x;
  for (int i = tid; i < 4096; i += 256) lds[i] = 1.0 + i * 1e-6;
  __syncthreads();
  double acc = 0;
  const long long t0 = __builtin_readcyclecounter();
  for (int p = 0; p < phases; ++p) {
    double v = 0;
#pragma unroll
    for (int l = 0; l < NLOAD; ++l) v += lds[(tid * 7 + l * 257 + p) & 4095];
#pragma unroll
    for (int d = 0; d < NDEP; ++d) v = fma(v, 1.0000001, 1e-9);
    lds[(tid + p) & 4095] = v;
    acc += v;
    __syncthreads();
  }
  const long long t1 = __builtin_readcyclecounter();
  if (tid == 0) out[blockIdx.x] = t1 - t0;
  sink[blockIdx.x * 256 + tid] = acc;
}
template <int NDEP, int NLOAD>
void run(long long *out, double *sink) {
  const int phases = 2000;
  for (int per_cu : {1, 2}) {
    const size_t lds = per_cu == 1 ? 100 * 1024 : 70 * 1024;     // forces 1 or 2 workgroups per CU
    hipFuncSetAttribute((const void *)k<NDEP, NLOAD>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((k<NDEP, NLOAD>), dim3(256 * per_cu), dim3(256), lds, 0, out, sink, phases, 0);
    hipDeviceSynchronize();
    long long h[8];
    hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
    printf("loads %2d  dependent ops %2d  workgroups/CU %d: %.0f cycles per phase\n", NLOAD, NDEP, per_cu, (double)h[0] / phases);
  }
}
int main() {
  long long *out; double *sink;
  hipMalloc(&out, 4096 * sizeof(long long)); hipMalloc(&sink, 8 << 20);
  run<0, 1>(out, sink); run<4, 1>(out, sink); run<10, 1>(out, sink); run<0, 8>(out, sink); run<10, 8>(out, sink); run<10, 20>(out, sink);
  return 0;
}
