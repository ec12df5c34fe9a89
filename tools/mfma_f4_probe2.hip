// mfma_f4_probe2.hip -- what do independent VALU instructions cost inside a stream of v_mfma_scale_f32_32x32x64_f8f6f4 (FP4), one wave per SIMD?
// 16 independent accumulators; after every MFMA, V integer VALU (v_and_b32 / v_lshrrev_b32 on registers no MFMA reads).
// Last variant: the VALU results ARE later MFMA operands (rewritten after every 4th MFMA), as in k_crossprod_f4.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef int v4i __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));
__device__ __forceinline__ v16f mfma_f4(v4i a, v4i b, v16f c) {
  v8i a8 = {a[0], a[1], a[2], a[3], 0, 0, 0, 0}, b8 = {b[0], b[1], b[2], b[3], 0, 0, 0, 0};
  return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, c, 4, 4, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
}
template <int V, bool DEP>
__global__ void __launch_bounds__(256, 1) k(v4i a0, v4i b0, float *sink, int iters, unsigned long long *cyc) {
  v16f acc[16];
  for (int t = 0; t < 16; t++) for (int r = 0; r < 16; r++) acc[t][r] = 0.f;
  v4i a = a0, b = b0;
  a[0] += threadIdx.x & 1;
  uint32_t x0 = threadIdx.x * 2654435761u, x1 = x0 ^ 0x9e3779b9u, x2 = x0 + 77u, x3 = x1 + 13u;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int t = 0; t < 16; t++) {
      acc[t] = mfma_f4(a, b, acc[t]);
#pragma unroll
      for (int v = 0; v < V; v++) {
        if (v % 4 == 0) asm volatile("v_and_b32 %0, 0x33333333, %1" : "=v"(x0) : "v"(x1));
        if (v % 4 == 1) asm volatile("v_lshrrev_b32 %0, 2, %1" : "=v"(x1) : "v"(x2));
        if (v % 4 == 2) asm volatile("v_and_b32 %0, 0x33333333, %1" : "=v"(x2) : "v"(x3));
        if (v % 4 == 3) asm volatile("v_lshrrev_b32 %0, 2, %1" : "=v"(x3) : "v"(x0));
      }
      if (DEP && (t & 3) == 3) { a[t >> 2] = (int)(x0 & 0x33333333u); b[t >> 2] = (int)(x2 & 0x33333333u); }
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  for (int t = 0; t < 16; t++) for (int r = 0; r < 16; r++) s += acc[t][r];
  if (s == 12345.f) sink[0] = s + x0 + x1 + x2 + x3;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int V, bool DEP>
void run(const char *name, unsigned long long *dc, float *sink) {
  v4i a = {0x12121212, 0x21212121, 0x11111111, 0x22222222}, b = {0x21212121, 0x12121212, 0x22222222, 0x11111111};
  const int iters = 2000, grid = 256;
  hipLaunchKernelGGL((k<V, DEP>), dim3(grid), dim3(256), 0, 0, a, b, sink, 50, dc);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL((k<V, DEP>), dim3(grid), dim3(256), 0, 0, a, b, sink, iters, dc);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> hc(grid);
  (void)hipMemcpy(hc.data(), dc, 8 * grid, hipMemcpyDeviceToHost);
  printf("F4PROBE2 %-28s V=%d: %.2f Pop/s, %.1f shader cycles per MFMA (+%d VALU)\n", name, V, grid * 4.0 * iters * 16 * 2.0 * 32 * 32 * 64 / (ms * 1e-3) * 1e-15,
         (double)hc[0] / (iters * 16.0), V);
}
int main() {
  unsigned long long *dc; float *sink;
  (void)hipMalloc(&dc, 8 * 2048); (void)hipMalloc(&sink, 4);
  run<0, false>("independent VALU", dc, sink); run<1, false>("independent VALU", dc, sink); run<2, false>("independent VALU", dc, sink);
  run<3, false>("independent VALU", dc, sink); run<4, false>("independent VALU", dc, sink); run<6, false>("independent VALU", dc, sink);
  run<8, false>("independent VALU", dc, sink);
  run<3, true>("operands rewritten each 4", dc, sink);
  return 0;
}
