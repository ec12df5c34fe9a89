// int8 MFMA issue rates on gfx950: v_mfma_i32_32x32x32_i8 and v_mfma_i32_16x16x64_i8, 1 wave per SIMD, random operands, in-kernel clock
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

template <int SHAPE, int NACC>
__global__ void __launch_bounds__(256, 1) k(const int *src, int *out, int iters, unsigned long long *clk) {
  v4i a[4], b[4];
  for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) { a[i][j] = src[(threadIdx.x * 16 + i * 4 + j) & 4095] & 0x03030303; b[i][j] = src[(threadIdx.x * 16 + i * 4 + j + 7) & 4095] & 0x03030303; }
  unsigned long long t0, r0, t1, r1;
  int s = 0;
  if (SHAPE == 32) {
    v16i acc[NACC];
    for (int i = 0; i < NACC; i++) for (int r = 0; r < 16; r++) acc[i][r] = 0;
    t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; it++) {
#pragma unroll
      for (int i = 0; i < NACC; i++) acc[i] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[i & 3], b[(i >> 2) & 3], acc[i], 0, 0, 0);
    }
    t1 = __builtin_amdgcn_s_memtime(); r1 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < NACC; i++) for (int r = 0; r < 16; r++) s += acc[i][r];
  } else {
    v4i acc[NACC];
    for (int i = 0; i < NACC; i++) for (int r = 0; r < 4; r++) acc[i][r] = 0;
    t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; it++) {
#pragma unroll
      for (int i = 0; i < NACC; i++) acc[i] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[i & 3], b[(i >> 2) & 3], acc[i], 0, 0, 0);
    }
    t1 = __builtin_amdgcn_s_memtime(); r1 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < NACC; i++) for (int r = 0; r < 4; r++) s += acc[i][r];
  }
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
  auto run = [&](const char *name, auto launch, double ops_per_inst, int nacc) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    launch(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0)); for (int i = 0; i < 3; i++) launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 3;
    std::vector<unsigned long long> hc(2 * ncu); CK(hipMemcpy(hc.data(), clk, 16 * ncu, hipMemcpyDeviceToHost));
    double ghz = (double)hc[0] / (double)hc[1] * 0.1;
    double ops = (double)ncu * 4 * iters * nacc * ops_per_inst;
    printf("%-34s %8.3f ms  %6.2f Pop/s  clock %.3f GHz  cycles/MFMA %.2f\n", name, ms, ops / ms * 1e-12, ghz, (double)hc[0] / ((double)iters * nacc));
  };
  run("i8 32x32x32, 16 acc, 1 wave/SIMD", [&] { k<32, 16><<<ncu, 256>>>(src, out, iters, clk); }, 2.0 * 32 * 32 * 32, 16);
  run("i8 32x32x32, 4 acc, 1 wave/SIMD", [&] { k<32, 4><<<ncu, 256>>>(src, out, iters, clk); }, 2.0 * 32 * 32 * 32, 4);
  run("i8 16x16x64, 16 acc, 1 wave/SIMD", [&] { k<16, 16><<<ncu, 256>>>(src, out, iters, clk); }, 2.0 * 16 * 16 * 64, 16);
  run("i8 16x16x64, 64 acc, 1 wave/SIMD", [&] { k<16, 64><<<ncu, 256>>>(src, out, iters, clk); }, 2.0 * 16 * 16 * 64, 64);
  return 0;
}
