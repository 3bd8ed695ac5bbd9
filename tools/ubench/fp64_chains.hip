// Microbenchmark: v_fma_f64 issue interval vs dependent latency -- NCH independent accumulation chains per lane.
// build: hipcc --offload-arch=gfx950 -O3 fp64_chains.hip -o fp64_chains
#include <hip/hip_runtime.h>
#include <cstdio>
template <int NCH>
__global__ void k(long long *out, double *sink, int iters) {
  double f[NCH];
  const double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-6;
#pragma unroll
  for (int c = 0; c < NCH; ++c) f[c] = a + c;
  __syncthreads();
  const long long t0 = __builtin_readcyclecounter();
  constexpr int U = 64 / NCH > 0 ? 64 / NCH : 1;   // >= 64 FMAs per loop trip, so the loop branch does not dominate
  for (int i = 0; i < iters / U; ++i) {
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int c = 0; c < NCH; ++c) f[c] = fma(f[c], b, a);
  }
  const long long t1 = __builtin_readcyclecounter();
  double s = 0;
#pragma unroll
  for (int c = 0; c < NCH; ++c) s += f[c];
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
  sink[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NCH>
void run(long long *out, double *sink) {
  const int iters = 2048;
  for (int nw : {4, 8, 16}) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<NCH>, dim3(256), dim3(64 * nw), 0, 0, out, sink, iters);   // warm-up
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k<NCH>, dim3(256), dim3(64 * nw), 0, 0, out, sink, iters);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double tf = 2.0 * 256.0 * 64 * nw * NCH * (iters / (64 / NCH > 0 ? 64 / NCH : 1) * (64 / NCH > 0 ? 64 / NCH : 1)) / (ms * 1e-3) / 1e12;
    long long h[16];
    hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
    printf("chains %2d  waves/SIMD %d: cycles per FMA per wave %.2f  (per SIMD %.2f)  wall %.3f ms = %.1f TFLOP/s, counter %.2f GHz\n", NCH, nw / 4, (double)h[0] / (iters / (64 / NCH > 0 ? 64 / NCH : 1) * (64 / NCH > 0 ? 64 / NCH : 1)) / NCH, (double)h[0] / (iters / (64 / NCH > 0 ? 64 / NCH : 1) * (64 / NCH > 0 ? 64 / NCH : 1)) / NCH / (nw / 4), ms, tf, (double)h[0] / (ms * 1e-3) / 1e9);
  }
}
int main() {
  long long *out; double *sink;
  hipMalloc(&out, 8192 * sizeof(long long)); hipMalloc(&sink, 4 << 20);
  run<1>(out, sink); run<2>(out, sink); run<4>(out, sink); run<8>(out, sink); run<16>(out, sink); run<32>(out, sink);
  return 0;
}
