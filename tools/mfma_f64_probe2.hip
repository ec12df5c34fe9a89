// Empirical lane map of v_mfma_f64_4x4x4_4b_f64 on gfx950, incl. cbsz/abid A-broadcast.
// For every (laneA, laneB) one-hot pair records which D lanes become non-zero.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <int CBSZ, int ABID>
__global__ void k_onehot(double *out) {
  int pa = blockIdx.x >> 6, pb = blockIdx.x & 63, l = threadIdx.x;
  double a = (l == pa) ? 1.0 : 0.0, b = (l == pb) ? 1.0 : 0.0;
  double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, CBSZ, ABID, 0);
  out[(size_t)blockIdx.x * 64 + l] = d;
}

template <int CBSZ, int ABID>
void run(double *dout, std::vector<double> &h) {
  k_onehot<CBSZ, ABID><<<4096, 64>>>(dout);
  CK(hipMemcpy(h.data(), dout, sizeof(double) * 4096 * 64, hipMemcpyDeviceToHost));
  printf("=== cbsz=%d abid=%d: (laneA,laneB)->D lanes\n", CBSZ, ABID);
  // for each A lane: list of (B lane -> D lanes)
  for (int pa = 0; pa < 64; pa++) {
    printf("A%02d:", pa);
    for (int pb = 0; pb < 64; pb++) {
      bool any = false;
      for (int l = 0; l < 64; l++) if (h[((size_t)(pa * 64 + pb)) * 64 + l] != 0.0) { if (!any) printf(" B%02d>", pb); any = true; printf("D%02d,", l); }
    }
    printf("\n");
  }
}

template <int NACC, int NF>
__global__ void __launch_bounds__(256) k_mix4(double *out, int iters, double a0, double b0) {
  double acc[NACC];
  double f[NF > 0 ? NF : 1];
  for (int i = 0; i < NACC; i++) acc[i] = 0;
  for (int i = 0; i < NF; i++) f[i] = i;
  double a = a0 + threadIdx.x * 1e-9, b = b0;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < NACC; i += 4) {
      acc[i + 0] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[i + 0], 2, 0, 0);
      acc[i + 1] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[i + 1], 2, 1, 0);
      acc[i + 2] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[i + 2], 2, 2, 0);
      acc[i + 3] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[i + 3], 2, 3, 0);
      if (i < NF) { f[i] = __builtin_fma(a, f[i], b); f[i+1] = __builtin_fma(a, f[i+1], b); f[i+2] = __builtin_fma(a, f[i+2], b); f[i+3] = __builtin_fma(a, f[i+3], b); }
    }
  }
  double s = 0;
  for (int i = 0; i < NACC; i++) s += acc[i];
  for (int i = 0; i < NF; i++) s += f[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
  double *dout; CK(hipMalloc(&dout, sizeof(double) * 4096 * 64));
  std::vector<double> h(4096 * 64);
  run<0, 0>(dout, h);
  run<2, 0>(dout, h);
  run<2, 1>(dout, h);
  run<2, 3>(dout, h);
  run<1, 0>(dout, h);
  run<1, 1>(dout, h);
  run<1, 2>(dout, h);
  // throughput with broadcast + VALU mix
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  int ncu = p.multiProcessorCount; const int iters = 20000;
  double *out; CK(hipMalloc(&out, sizeof(double) * 256 * ncu * 16));
  auto timeit = [&](auto launch) { hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); launch(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0)); for (int i = 0; i < 5; i++) launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); return (double)ms / 5; };
#define RUN(NACC, NF, BPC) { int blocks = ncu * BPC; double ms = timeit([&] { k_mix4<NACC, NF><<<blocks, 256>>>(out, iters, 1.0, 0.5); }); \
  double fl = (double)blocks * 4 * iters * (512.0 * NACC + 128.0 * NF); printf("mfma4x4x4 bcast acc=%d fma=%d blocks/CU=%d: %.3f ms %.2f TFLOP/s (mfma-only %.2f)\n", NACC, NF, BPC, ms, fl / ms * 1e-9, (double)blocks * 4 * iters * 512.0 * NACC / ms * 1e-9); }
  RUN(8, 0, 1) RUN(16, 0, 1) RUN(32, 0, 1) RUN(16, 0, 2) RUN(16, 4, 1) RUN(16, 8, 1) RUN(16, 16, 1) RUN(16, 16, 2) RUN(32,8,1)
  return 0;
}
