// fp64 4x4x4 MFMA ceiling on RANDOM operands + in-kernel clock (s_memtime / s_memrealtime), 64 accumulators like k_gemm<8,8>
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <random>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <int KIND>  // 0: a,b random normal; 1: a in {0,1,2} random, b random normal; 2: all constant
__global__ void __launch_bounds__(256, 2) k(const double *ra, const double *rb, double *out, int iters, unsigned long long *clk) {
  double a[8], b[8], acc[8][8];
  for (int i = 0; i < 8; i++) { a[i] = ra[(i * 256 + threadIdx.x) % 4096]; b[i] = rb[(i * 256 + threadIdx.x + blockIdx.x) % 4096]; }
  if (KIND == 2) for (int i = 0; i < 8; i++) { a[i] = 1.0; b[i] = 0.5; }
  for (int g = 0; g < 8; g++) for (int h = 0; h < 8; h++) acc[g][h] = 0;
  unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int g = 0; g < 8; g++)
#pragma unroll
      for (int h = 0; h < 8; h++) acc[g][h] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[g], b[h], acc[g][h], 0, 0, 0);
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  double s = 0;
  for (int g = 0; g < 8; g++) for (int h = 0; h < 8; h++) s += acc[g][h];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

int main() {
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  int ncu = p.multiProcessorCount; const int iters = 40000;
  std::mt19937_64 rng(1); std::normal_distribution<double> nd;
  std::vector<double> ha(4096), hb(4096), hz(4096);
  for (auto &x : ha) x = nd(rng); for (auto &x : hb) x = nd(rng); for (auto &x : hz) x = (double)(rng() % 3);
  double *da, *db, *dz, *out; unsigned long long *clk;
  CK(hipMalloc(&da, 32768)); CK(hipMalloc(&db, 32768)); CK(hipMalloc(&dz, 32768)); CK(hipMalloc(&out, sizeof(double) * 256 * ncu * 2)); CK(hipMalloc(&clk, 16 * ncu * 2));
  CK(hipMemcpy(da, ha.data(), 32768, hipMemcpyHostToDevice)); CK(hipMemcpy(db, hb.data(), 32768, hipMemcpyHostToDevice)); CK(hipMemcpy(dz, hz.data(), 32768, hipMemcpyHostToDevice));
  auto run = [&](const char *name, auto launch) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    launch(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0)); for (int i = 0; i < 3; i++) launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 3;
    std::vector<unsigned long long> hc(2 * ncu * 2); CK(hipMemcpy(hc.data(), clk, 16 * ncu * 2, hipMemcpyDeviceToHost));
    double ghz = (double)hc[0] / (double)hc[1] * 0.1;
    double fl = (double)ncu * 2 * 4 * iters * 64 * 512.0;
    printf("%-44s %8.3f ms %7.2f TFLOP/s  in-kernel clock %.3f GHz  cycles/MFMA/SIMD %.2f\n", name, ms, fl / ms * 1e-9, ghz, (double)hc[0] / ((double)iters * 64 * 2));
  };
  run("random normal a, random normal b", [&] { k<0><<<ncu * 2, 256>>>(da, db, out, iters, clk); });
  run("a in {0,1,2}, random normal b", [&] { k<1><<<ncu * 2, 256>>>(dz, db, out, iters, clk); });
  run("constant a=1.0 b=0.5", [&] { k<2><<<ncu * 2, 256>>>(da, db, out, iters, clk); });
  run("random normal a, random normal b (again)", [&] { k<0><<<ncu * 2, 256>>>(da, db, out, iters, clk); });
  return 0;
}
