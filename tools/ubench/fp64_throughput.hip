// fp64 vector THROUGHPUT on gfx950 (what bounds the prep kernel): NCH independent v_fma_f64 chains per lane, W waves per SIMD on every CU,
// kernels long enough (>= 20 ms) that launch effects and the clock ramp do not matter.  Reports, per configuration: wall-clock TFLOP/s (HIP events), the shader
// clock the launch sustained (s_memtime cycles of one wave / wall time) and, from the two, the SIMD cycles one fp64 wave-instruction occupies.
//   hipcc --offload-arch=gfx950 -O3 fp64_throughput.hip -o fp64_throughput && ./fp64_throughput
#include <hip/hip_runtime.h>
#include <cstdio>
template <int NCH>
__global__ __launch_bounds__(256) void k(long long *out, double *sink, int iters) {
  double f[NCH];
  const double a = threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-12;
#pragma unroll
  for (int c = 0; c < NCH; ++c) f[c] = a + c;
  const long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 64 / NCH; ++u)
#pragma unroll
      for (int c = 0; c < NCH; ++c) f[c] = fma(f[c], b, a);
  }
  const long long t1 = __builtin_readcyclecounter();
  double s = 0;
#pragma unroll
  for (int c = 0; c < NCH; ++c) s += f[c];
  if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
  sink[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NCH>
void run(long long *out, double *sink, int cus) {
  for (int wps : {1, 2, 3, 4}) {
    const int blocks = cus * wps, iters = 400000 / wps;      // one 256-thread workgroup = one wave on each SIMD of a CU
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<NCH>, dim3(blocks), dim3(256), 0, 0, out, sink, 2000);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k<NCH>, dim3(blocks), dim3(256), 0, 0, out, sink, iters);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    long long h[4]; hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
    const double fmas_per_wave = 64.0 * iters, ghz = h[0] / (ms * 1e6);
    const double tf = 2.0 * 64 * fmas_per_wave * 4.0 * wps * cus / (ms * 1e-3) / 1e12;
    printf("chains %2d  waves/SIMD %d: %7.2f ms  %5.1f TFLOP/s  clock %.2f GHz  cycles per FMA: %.2f per wave, %.2f per SIMD\n", NCH, wps, ms, tf, ghz, h[0] / fmas_per_wave, h[0] / fmas_per_wave / wps);
  }
}
int main() {
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  long long *out; double *sink;
  hipMalloc(&out, 8192 * sizeof(long long)); hipMalloc(&sink, 8 << 20);
  printf("%s, %d CUs\n", p.name, p.multiProcessorCount);
  run<1>(out, sink, p.multiProcessorCount); run<4>(out, sink, p.multiProcessorCount); run<16>(out, sink, p.multiProcessorCount);
  return 0;
}
