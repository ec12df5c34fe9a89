// How do integer VALU writes of the MFMA A operand interact with a stream of v_mfma_f64_4x4x4_4b_f64 on gfx950?
// 64 accumulators (8 A fragments x 8 B fragments) per "K-step", like k_gemm<8,8>.  Patterns P0..P6 below.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
#define SB() __builtin_amdgcn_sched_barrier(0)
// one integer VALU op that the compiler cannot hoist, CSE or reorder against other asm volatile statements
#define VMOV(dst, src) asm volatile("v_mov_b32 %0, %1" : "=v"(dst) : "v"(src))
#define VBFE(dst, src, sh) asm volatile("v_bfe_u32 %0, %1, %2, 2" : "=v"(dst) : "v"(src), "v"(sh))
#define VLSHLADD(dst, src, k) asm volatile("v_lshl_add_u32 %0, %1, 20, %2" : "=v"(dst) : "v"(src), "v"(k))

template <int P>
__global__ void __launch_bounds__(256, 2) k(double *out, int iters, const unsigned *wsrc) {
  double acc[8][8], bf[8];
  unsigned w[8], hi[8];
  for (int g = 0; g < 8; g++) { w[g] = wsrc[(threadIdx.x + g * 64) & 1023]; hi[g] = 0x3FF00000u; bf[g] = 1.0 + 0.125 * g + threadIdx.x * 1e-6; }
  for (int g = 0; g < 8; g++) for (int h = 0; h < 8; h++) acc[g][h] = 0;
  const unsigned sh = 2 * ((threadIdx.x & 63) >> 2), kc = 0x3FE00000u;
  for (int it = 0; it < iters; it++) {
    if (P == 2) { for (int g = 0; g < 8; g++) VMOV(hi[g], w[g]); SB(); }
    if (P == 4) { for (int g = 0; g < 8; g++) { unsigned z, t; VBFE(z, w[g], sh); VLSHLADD(t, z, kc); VMOV(z, t); VMOV(hi[g], z); } SB(); }
#pragma unroll
    for (int g = 0; g < 8; g++) {
      if (P == 1) { VMOV(hi[g], w[g]); SB(); }
      if (P == 3) { unsigned z, t; VBFE(z, w[g], sh); VLSHLADD(t, z, kc); VMOV(z, t); VMOV(hi[g], z); SB(); }
      const double af = __hiloint2double((int)hi[g], 0);
#pragma unroll
      for (int h = 0; h < 8; h++) {
        acc[g][h] = __builtin_amdgcn_mfma_f64_4x4x4f64(af, bf[h], acc[g][h], 0, 0, 0);
        if (P == 5 && (h & 1)) { SB(); unsigned t; VMOV(t, w[(g + 4) & 7]); w[(g + 4) & 7] = t; SB(); }   // 4 VALU per group, not feeding the next MFMAs
        if (P == 6 && (h & 1)) { SB(); VMOV(hi[(g + 4) & 7], w[(g + 4) & 7]); SB(); }                         // 4 VALU per group writing an A reg used 4 groups later
      }
      SB();
    }
  }
  double s = 0;
  for (int g = 0; g < 8; g++) for (int h = 0; h < 8; h++) s += acc[g][h];
  for (int g = 0; g < 8; g++) s += w[g] + hi[g];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  int ncu = p.multiProcessorCount; const int iters = 6000;
  double *out; unsigned *wsrc; CK(hipMalloc(&out, sizeof(double) * 256 * ncu * 2)); CK(hipMalloc(&wsrc, 4096)); CK(hipMemset(wsrc, 0x5A, 4096));
  auto run = [&](const char *name, auto launch, int bpc) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    launch(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0)); for (int i = 0; i < 3; i++) launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 3;
    double fl = (double)ncu * bpc * 4 * iters * 64 * 512.0;
    double cyc = ms * 1e-3 * 2.39e9 / ((double)iters * bpc);   // SIMD cycles per 64-MFMA K-step per wave slot
    printf("%-78s blocks/CU=%d %8.3f ms %7.2f TFLOP/s  (%.0f cyc per 64 MFMA, ideal 1024)\n", name, bpc, ms, fl / ms * 1e-9, cyc);
  };
#define RUN(P, NAME) for (int bpc = 1; bpc <= 2; bpc++) run(NAME, [&] { k<P><<<ncu * bpc, 256>>>(out, iters, wsrc); }, bpc);
  RUN(0, "P0: 64 MFMA only")
  RUN(1, "P1: 1 v_mov writes A reg immediately before its 8 MFMAs (8 VALU)")
  RUN(2, "P2: 8 v_mov clustered, then 64 MFMA (8 VALU)")
  RUN(3, "P3: 4 dependent VALU write A reg immediately before its 8 MFMAs (32 VALU)")
  RUN(4, "P4: 32 VALU clustered, then 64 MFMA")
  RUN(5, "P5: 32 VALU spread (1 per 2 MFMA), not feeding MFMAs")
  RUN(6, "P6: 32 VALU spread (1 per 2 MFMA), writing an A reg used 32 MFMAs later")
  return 0;
}
