// Device check of csrc/gelsd43.h (TEST ONLY): runs solve_ones on the GPU over matrices read from stdin-named binary files and writes x (and the
// stage records) back.  usage: gelsd43_device <A.bin (n x 12 float32)> <n> <x.bin out (n x 3)> <dbg.bin out (n x 64)>
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "gelsd43.h"

__global__ void k(int n, const float *A, float *x, float *dbg) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float xo[3], d[64];
  for (int j = 0; j < 64; ++j) d[j] = 0.f;
  mpc::gelsd43::solve_ones(A + 12 * i, xo, d);
  for (int j = 0; j < 3; ++j) x[3 * i + j] = xo[j];
  for (int j = 0; j < 64; ++j) dbg[64 * i + j] = d[j];
}
__global__ void k_nodbg(int n, const float *A, float *x) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float xo[3];
  mpc::gelsd43::solve_ones(A + 12 * i, xo);
  for (int j = 0; j < 3; ++j) x[3 * i + j] = xo[j];
}
int main(int argc, char **argv) {
  if (argc < 5) return 2;
  const int n = atoi(argv[2]);
  std::vector<float> A(12 * (size_t)n), x(3 * (size_t)n), x2(3 * (size_t)n), dbg(64 * (size_t)n);
  FILE *f = fopen(argv[1], "rb");
  if (!f || fread(A.data(), 4, A.size(), f) != A.size()) return 3;
  fclose(f);
  float *dA, *dx, *dd;
  if (hipMalloc(&dA, A.size() * 4) != hipSuccess || hipMalloc(&dx, x.size() * 4) != hipSuccess || hipMalloc(&dd, dbg.size() * 4) != hipSuccess) return 4;
  hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3((n + 63) / 64), dim3(64), 0, 0, n, dA, dx, dd);
  hipMemcpy(x.data(), dx, x.size() * 4, hipMemcpyDeviceToHost);
  hipMemcpy(dbg.data(), dd, dbg.size() * 4, hipMemcpyDeviceToHost);
  hipLaunchKernelGGL(k_nodbg, dim3((n + 63) / 64), dim3(64), 0, 0, n, dA, dx);
  hipMemcpy(x2.data(), dx, x2.size() * 4, hipMemcpyDeviceToHost);
  if (hipDeviceSynchronize() != hipSuccess) return 5;
  int diff = 0;
  for (size_t i = 0; i < x.size(); ++i) diff += x[i] != x2[i];
  printf("with / without the stage record: %d entries differ\n", diff);
  f = fopen(argv[3], "wb"); fwrite(x2.data(), 4, x2.size(), f); fclose(f);
  f = fopen(argv[4], "wb"); fwrite(dbg.data(), 4, dbg.size(), f); fclose(f);
  return 0;
}
