// Microbenchmark: issue rate of v_mfma_f64_16x16x4_f64 vs v_fma_f64 on one SIMD, alone and co-scheduled.
// build: hipcc --offload-arch=gfx950 -O3 fp64_rates.hip -o fp64_rates
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
__global__ void k(long long *out, double *sink, int mode, int iters) {
  const int wave = threadIdx.x / 64;
  d4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
  double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-6;
  double f0 = a, f1 = a + 1, f2 = a + 2, f3 = a + 3, f4 = a + 4, f5 = a + 5, f6 = a + 6, f7 = a + 7;
  __syncthreads();
  long long t0 = __builtin_readcyclecounter();
  const bool do_mfma = mode == 0 || (mode == 2 && (wave & 4) == 0);
  const bool do_fma = mode == 1 || (mode == 2 && (wave & 4) != 0);
  if (do_mfma) {
    for (int i = 0; i < iters; ++i) {
      c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
    }
  }
  if (do_fma) {
    for (int i = 0; i < iters; ++i) {
      f0 = fma(f0, b, a); f1 = fma(f1, b, a); f2 = fma(f2, b, a); f3 = fma(f3, b, a);
      f4 = fma(f4, b, a); f5 = fma(f5, b, a); f6 = fma(f6, b, a); f7 = fma(f7, b, a);
    }
  }
  long long t1 = __builtin_readcyclecounter();
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * (blockDim.x / 64) + wave] = t1 - t0;
  sink[blockIdx.x * blockDim.x + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3] + f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7;
}
int main() {
  long long *out; double *sink;
  hipMalloc(&out, 4096 * sizeof(long long)); hipMalloc(&sink, 1 << 20);
  const int iters = 2000;
  for (int mode = 0; mode < 3; ++mode)
    for (int nw : {4, 8}) {
      hipLaunchKernelGGL(k, dim3(256), dim3(64 * nw), 0, 0, out, sink, mode, iters);
      hipDeviceSynchronize();
      long long h[16]; hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
      const char *nm[] = {"mfma_f64_16x16x4 (4/iter)", "v_fma_f64 (8/iter)", "waves 0-3 mfma + waves 4-7 fma"};
      printf("%-34s waves/block %d: cycles/iter wave0 %.1f  wave%d %.1f\n", nm[mode], nw, (double)h[0] / iters, nw - 1, (double)h[nw - 1] / iters);
    }
  return 0;
}
