// Is the rate / clock of v_mfma_f64_4x4x4_4b_f64 operand-dependent?  Bare loop, 2 waves per SIMD, 32 accumulators per wave, operands in registers:
// "den" = the genotype operand of k_gemm (low word 0..2, high word 0: z * 2^-1074), "rnd" = scaled random doubles near 2^900 (the B operand).
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_f64_probe7 mfma_f64_probe7.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__global__ void __launch_bounds__(256, 2) k(const double *srcA, const double *srcB, double *out, int iters, unsigned long long *clk) {
  double a[8], b[8], acc[32];
  for (int i = 0; i < 8; i++) { a[i] = srcA[(threadIdx.x * 8 + i + blockIdx.x * 97) & 65535]; b[i] = srcB[(threadIdx.x * 8 + i + 7 + blockIdx.x * 31) & 65535]; }
  for (int i = 0; i < 32; i++) acc[i] = 0.0;
  unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < 32; i++) acc[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[i & 7], b[(i >> 2) & 7], acc[i], 0, 0, 0);
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  double s = 0;
  for (int i = 0; i < 32; i++) s += acc[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

int main() {
  const int blocks = 256 * 2, iters = 40000;
  std::vector<double> den(65536), rnd(65536);
  srand(1);
  for (int i = 0; i < 65536; i++) {
    unsigned long long z = rand() % 3; double d; memcpy(&d, &z, 8); den[i] = d;
    rnd[i] = ldexp(((double)rand() / RAND_MAX - 0.5) + 1e-3 * ((double)rand() / RAND_MAX), 899);
  }
  double *dden, *drnd, *out; unsigned long long *clk;
  CK(hipMalloc(&dden, 65536 * 8)); CK(hipMalloc(&drnd, 65536 * 8)); CK(hipMalloc(&out, blocks * 256 * 8)); CK(hipMalloc(&clk, blocks * 16));
  CK(hipMemcpy(dden, den.data(), 65536 * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(drnd, rnd.data(), 65536 * 8, hipMemcpyHostToDevice));
  const char *names[4] = {"A den, B den", "A den, B rnd (k_gemm today)", "A rnd, B den (swapped)", "A rnd, B rnd"};
  for (int rep = 0; rep < 2; rep++)
    for (int c = 0; c < 4; c++) {
      const double *pa = (c & 2) ? drnd : dden, *pb = (c & 1) ? drnd : dden;
      hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
      hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, pa, pb, out, iters / 4, clk);
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, pa, pb, out, iters, clk);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      std::vector<unsigned long long> h(2 * blocks);
      CK(hipMemcpy(h.data(), clk, blocks * 16, hipMemcpyDeviceToHost));
      const double ghz = (double)h[0] / ((double)h[1] * 10.0);
      const double flop = 512.0 * 32.0 * iters * 4.0 * blocks;
      printf("%-30s: %.2f ms, %.2f TFLOP/s, in-kernel clock %.3f GHz\n", names[c], ms, flop / (ms * 1e-3) * 1e-12, ghz);
    }
  return 0;
}
