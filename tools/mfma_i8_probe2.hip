// How many integer VALU ops hide behind v_mfma_i32_32x32x32_i8 with ONE wave per SIMD (the k_crossprod2 regime)?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
#define SB() __builtin_amdgcn_sched_barrier(0)
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
#define VOP(d, s) asm volatile("v_and_b32 %0, 0x03030303, %1" : "=v"(d) : "v"(s))

// PER = VALU per MFMA placed right after each MFMA; CL = VALU clustered before each group of 4 MFMAs
template <int PER, int CL>
__global__ void __launch_bounds__(256, 1) k(const int *src, int *out, int iters, unsigned long long *clk) {
  v4i a[4], b[4];
  int w[8];
  for (int i = 0; i < 8; i++) w[i] = src[(threadIdx.x + i * 64) & 4095];
  for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) { a[i][j] = src[(threadIdx.x * 16 + i * 4 + j) & 4095] & 0x03030303; b[i][j] = src[(threadIdx.x * 16 + i * 4 + j + 7) & 4095] & 0x03030303; }
  v16i acc[16];
  for (int i = 0; i < 16; i++) for (int r = 0; r < 16; r++) acc[i][r] = 0;
  unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int g = 0; g < 4; g++) {
#pragma unroll
      for (int c = 0; c < CL; c++) { int t; VOP(t, w[c & 7]); w[(c + 1) & 7] = t; }
      SB();
#pragma unroll
      for (int h = 0; h < 4; h++) {
        acc[g * 4 + h] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[g], b[h], acc[g * 4 + h], 0, 0, 0);
        SB();
#pragma unroll
        for (int c = 0; c < PER; c++) { int t; VOP(t, w[(c + h) & 7]); w[(c + h + 1) & 7] = t; }
        SB();
      }
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  int s = 0;
  for (int i = 0; i < 16; i++) for (int r = 0; r < 16; r++) s += acc[i][r];
  for (int i = 0; i < 8; i++) s += w[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

int main() {
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  int ncu = p.multiProcessorCount; const int iters = 20000;
  std::vector<int> h(4096); for (auto &x : h) x = rand();
  int *src, *out; unsigned long long *clk;
  CK(hipMalloc(&src, 16384)); CK(hipMalloc(&out, 4 * 256 * ncu)); CK(hipMalloc(&clk, 16 * ncu));
  CK(hipMemcpy(src, h.data(), 16384, hipMemcpyHostToDevice));
  auto run = [&](const char *name, auto launch) {
    launch(); CK(hipDeviceSynchronize()); launch(); CK(hipDeviceSynchronize());
    std::vector<unsigned long long> hc(2 * ncu); CK(hipMemcpy(hc.data(), clk, 16 * ncu, hipMemcpyDeviceToHost));
    printf("%-52s clock %.3f GHz  cycles per 16 MFMA %.0f (ideal 512)\n", name, (double)hc[0] / (double)hc[1] * 0.1, (double)hc[0] / iters);
  };
  run("0 VALU", [&] { k<0, 0><<<ncu, 256>>>(src, out, iters, clk); });
  run("2 VALU after each MFMA (32 / K-step)", [&] { k<2, 0><<<ncu, 256>>>(src, out, iters, clk); });
  run("4 VALU after each MFMA (64 / K-step)", [&] { k<4, 0><<<ncu, 256>>>(src, out, iters, clk); });
  run("6 VALU after each MFMA (96 / K-step)", [&] { k<6, 0><<<ncu, 256>>>(src, out, iters, clk); });
  run("8 VALU clustered before each 4 MFMA (32 / K-step)", [&] { k<0, 8><<<ncu, 256>>>(src, out, iters, clk); });
  run("14 VALU clustered before each 4 MFMA (56 / K-step)", [&] { k<0, 14><<<ncu, 256>>>(src, out, iters, clk); });
  run("28 VALU clustered before each 4 MFMA (112 / K-step)", [&] { k<0, 28><<<ncu, 256>>>(src, out, iters, clk); });
  return 0;
}
