// Shader clock under an fp64 FMA load: cycles of __builtin_readcyclecounter (s_memtime) per second of HIP-event time,
// with 1 / 2 / 4 waves per SIMD busy on every CU.   hipcc --offload-arch=gfx950 -O3 clock_rate.hip -o clock_rate
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void spin(long long *out, int iters) {
  double a = threadIdx.x * 1e-3 + 1.0, b = 1.0000001, c = 1e-9;
  const long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int k = 0; k < 32; ++k) a = a * b + c;
  }
  const long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0 + (a == 0.5 ? 1 : 0);
}
int main() {
  long long *out; hipMalloc(&out, sizeof(long long) * 4096);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int wps : {1, 2, 4}) {
    const int blocks = 256 * wps, iters = 20000;
    spin<<<blocks, 256>>>(out, 1000);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    spin<<<blocks, 256>>>(out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long h[8]; hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
    printf("waves/SIMD %d: %.3f ms, %lld counter cycles in block 0 -> %.3f GHz; %.2f cycles per FMA per wave\n", wps, ms, h[0], h[0] / (ms * 1e6), (double)h[0] / (iters * 32.0));
  }
  return 0;
}
