// Is the power-bound rate of v_mfma_i32_32x32x32_i8 operand-dependent?  Bare MFMA loop (one wave per SIMD, 16 accumulators, operands in
// registers) with LOW-entropy data (genotype-like bytes 0..2) or FULL-entropy bytes (radix-256 digits) in the A and / or the B operand.
// Prints executed Pop/s and the in-kernel clock for the four combinations.  build: hipcc --offload-arch=gfx950 -O3 -o mfma_i8_probe3 mfma_i8_probe3.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

__global__ void __launch_bounds__(256, 1) k(const int *srcA, const int *srcB, int *out, int iters, unsigned long long *clk) {
  v4i a[4], b[4];
  for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) {
    a[i][j] = srcA[(threadIdx.x * 16 + i * 4 + j + blockIdx.x * 97) & 65535];
    b[i][j] = srcB[(threadIdx.x * 16 + i * 4 + j + 7 + blockIdx.x * 31) & 65535];
  }
  v16i acc[16];
  for (int i = 0; i < 16; i++) for (int r = 0; r < 16; r++) acc[i][r] = 0;
  unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int g = 0; g < 4; g++)
#pragma unroll
      for (int h = 0; h < 4; h++) acc[g * 4 + h] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[g], b[h], acc[g * 4 + h], 0, 0, 0);
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  int s = 0;
  for (int i = 0; i < 16; i++) for (int r = 0; r < 16; r++) s += acc[i][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

int main() {
  const int blocks = 256 * 4, iters = 20000;
  std::vector<int> lo(65536), hi(65536);
  srand(1);
  for (int i = 0; i < 65536; i++) {
    unsigned l = 0, h = 0;
    for (int b = 0; b < 4; b++) { l |= (unsigned)(rand() % 3) << (8 * b); h |= (unsigned)(rand() & 255) << (8 * b); }
    lo[i] = (int)l; hi[i] = (int)h;
  }
  int *dlo, *dhi, *out; unsigned long long *clk;
  CK(hipMalloc(&dlo, 65536 * 4)); CK(hipMalloc(&dhi, 65536 * 4)); CK(hipMalloc(&out, blocks * 256 * 4)); CK(hipMalloc(&clk, blocks * 16));
  CK(hipMemcpy(dlo, lo.data(), 65536 * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dhi, hi.data(), 65536 * 4, hipMemcpyHostToDevice));
  const char *names[4] = {"A low  (0..2), B low ", "A low  (0..2), B full", "A full, B low  (0..2)", "A full, B full       "};
  for (int rep = 0; rep < 2; rep++)
    for (int c = 0; c < 4; c++) {
      const int *pa = (c & 2) ? dhi : dlo, *pb = (c & 1) ? dhi : dlo;
      hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
      hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, pa, pb, out, iters / 10, clk);   // warm
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, pa, pb, out, iters, clk);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      std::vector<unsigned long long> h(2 * blocks);
      CK(hipMemcpy(h.data(), clk, blocks * 16, hipMemcpyDeviceToHost));
      double ghz = (double)h[0] / ((double)h[1] * 10.0);
      const double ops = 2.0 * 32 * 32 * 32 * 16.0 * iters * 4.0 * blocks;
      printf("%s: %.2f ms, %.3f Pop/s, in-kernel clock %.3f GHz\n", names[c], ms, ops / (ms * 1e-3) * 1e-15, ghz);
    }
  return 0;
}
