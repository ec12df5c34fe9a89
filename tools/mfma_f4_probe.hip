// mfma_f4_probe.hip -- v_mfma_scale_f32_32x32x64_f8f6f4 with FP4 (e2m1) operands as an EXACT small-integer engine for the crossproduct.
// A 2-bit field z in {0,1,2,3} placed in the low bits of a nibble IS the e2m1 number z/2 (0000 = 0, 0001 = 0.5, 0010 = 1, 0011 = 1.5),
// products are multiples of 1/4 and the fp32 accumulator is exact while 4 * sum < 2^24.
//  (1) exactness + operand lane map: D = A * B^T for random z against a host integer product, operands laid out as
//      lane l = (row l&31, K half l>>5), nibble j of the lane's 16 bytes = k = 32*(l>>5) + j;
//  (2) accumulation up to 2^24 quarter-units stays exact (K = 1.8 M with all z = 3);
//  (3) rate: cycles per instruction with 16 independent accumulators, one wave per SIMD and two.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef int v4i __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));

__device__ __forceinline__ v16f mfma_f4(v4i a, v4i b, v16f c) {
  v8i a8 = {a[0], a[1], a[2], a[3], 0, 0, 0, 0}, b8 = {b[0], b[1], b[2], b[3], 0, 0, 0, 0};
  return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, c, 4, 4, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
}

__global__ void k_exact(const v4i *A, const v4i *B, float *D, int reps) {
  v16f acc;
  for (int r = 0; r < 16; r++) acc[r] = 0.f;
  const v4i a = A[threadIdx.x], b = B[threadIdx.x];
  for (int i = 0; i < reps; i++) acc = mfma_f4(a, b, acc);
  for (int r = 0; r < 16; r++) D[threadIdx.x * 16 + r] = acc[r];
}

__global__ void __launch_bounds__(256) k_rate(v4i a0, v4i b0, float *sink, int iters, unsigned long long *cyc) {
  v16f acc[16];
  for (int t = 0; t < 16; t++) for (int r = 0; r < 16; r++) acc[t][r] = 0.f;
  v4i a = a0, b = b0;
  a[0] += threadIdx.x & 1;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int t = 0; t < 16; t++) acc[t] = mfma_f4(a, b, acc[t]);
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  for (int t = 0; t < 16; t++) for (int r = 0; r < 16; r++) s += acc[t][r];
  if (s == 12345.f) sink[0] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

int main() {
  // (1) exactness
  std::vector<uint8_t> za(32 * 64), zb(32 * 64);
  srand(7);
  for (auto &v : za) v = rand() & 3;
  for (auto &v : zb) v = rand() & 3;
  std::vector<uint32_t> ha(64 * 4, 0), hb(64 * 4, 0);
  for (int l = 0; l < 64; l++)
    for (int j = 0; j < 32; j++) {
      const int row = l & 31, k = 32 * (l >> 5) + j;
      ha[l * 4 + j / 8] |= (uint32_t)za[row * 64 + k] << (4 * (j % 8));
      hb[l * 4 + j / 8] |= (uint32_t)zb[row * 64 + k] << (4 * (j % 8));
    }
  v4i *dA, *dB; float *dD;
  hipMalloc(&dA, 1024); hipMalloc(&dB, 1024); hipMalloc(&dD, 64 * 16 * 4);
  hipMemcpy(dA, ha.data(), 1024, hipMemcpyHostToDevice); hipMemcpy(dB, hb.data(), 1024, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k_exact, dim3(1), dim3(64), 0, 0, dA, dB, dD, 1);
  std::vector<float> hd(1024);
  hipMemcpy(hd.data(), dD, 4096, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int l = 0; l < 64; l++)
    for (int r = 0; r < 16; r++) {
      const int col = l & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);   // D[row = A row][col = B row]
      int s = 0;
      for (int k = 0; k < 64; k++) s += za[row * 64 + k] * zb[col * 64 + k];
      if (hd[l * 16 + r] != 0.25f * s) { if (bad < 5) printf("mismatch lane %d reg %d: got %g want %g\n", l, r, hd[l * 16 + r], 0.25f * s); bad++; }
    }
  printf("F4PROBE exactness (D = A B^T / 4, standard 32x32 C/D map, k = 32*(lane>>5) + nibble): %s (%d mismatches)\n", bad ? "FAILED" : "ok", bad);
  // (2) long accumulation: all z = 3 -> each instruction adds 64 * 9 / 4 = 144; 29 000 reps -> 4 176 000 = 0.996 * 2^22 ... go to 2^24 / 4 in value
  for (auto &v : ha) v = 0x33333333u;
  hipMemcpy(dA, ha.data(), 1024, hipMemcpyHostToDevice); hipMemcpy(dB, ha.data(), 1024, hipMemcpyHostToDevice);
  const int reps = 29127;   // 29127 * 144 = 4 194 288 < 2^22 = 4 194 304 (value), i.e. 16 777 152 quarter units < 2^24
  hipLaunchKernelGGL(k_exact, dim3(1), dim3(64), 0, 0, dA, dB, dD, reps);
  hipMemcpy(hd.data(), dD, 4096, hipMemcpyDeviceToHost);
  bad = 0;
  for (int i = 0; i < 1024; i++) if (hd[i] != 144.0f * reps) bad++;
  printf("F4PROBE accumulation of %d instructions of all-3 operands (K = %d): %s (value %.1f, want %.1f)\n", reps, reps * 64, bad ? "FAILED" : "exact", hd[0], 144.0 * reps);
  // (3) rate
  unsigned long long *dc; float *sink;
  hipMalloc(&dc, 8 * 2048); hipMalloc(&sink, 4);
  v4i a = {0x12121212, 0x21212121, 0x11111111, 0x22222222}, b = {0x21212121, 0x12121212, 0x22222222, 0x11111111};
  for (int wg_threads : {256, 512}) {
    hipFuncSetAttribute((const void *)k_rate, hipFuncAttributeMaxDynamicSharedMemorySize, 0);
    const int iters = 4000, grid = 256;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k_rate, dim3(grid), dim3(wg_threads), 0, 0, a, b, sink, 100, dc);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_rate, dim3(grid), dim3(wg_threads), 0, 0, a, b, sink, iters, dc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> hc(grid);
    hipMemcpy(hc.data(), dc, 8 * grid, hipMemcpyDeviceToHost);
    const double waves = (double)grid * wg_threads / 64, inst = waves * iters * 16.0;
    printf("F4PROBE rate, %d waves per SIMD: %.3f ms, %.2f Pflop/s (2*32*32*64 per instruction), shader cycles per instruction per wave %.1f\n", wg_threads / 256, ms,
           inst * 2.0 * 32 * 32 * 64 / (ms * 1e-3) * 1e-15, (double)hc[0] / (iters * 16.0));
  }
  return 0;
}
