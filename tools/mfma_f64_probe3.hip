// How many integer VALU ops / LDS reads fit beside each v_mfma_f64_4x4x4_4b_f64 (16-cycle) on gfx950?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

// per iteration: 32 MFMAs (32 accumulators), NI*32/8... integer ops spread evenly, NL ds_read_b64
template <int NI, int NL>
__global__ void __launch_bounds__(256) k(double *out, int iters, double a0, double b0, unsigned seed) {
  __shared__ double lds[2048];
  for (int i = threadIdx.x; i < 2048; i += 256) lds[i] = i * 0.5;
  __syncthreads();
  double acc[32];
  for (int i = 0; i < 32; i++) acc[i] = 0;
  double a = a0 + threadIdx.x * 1e-9, b = b0;
  unsigned x[8];
  for (int i = 0; i < 8; i++) x[i] = seed + threadIdx.x * 7 + i;
  int lo = threadIdx.x & 63;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < 32; i++) {
      acc[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[i], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < NI; j++) { x[(i + j) & 7] = (x[(i + j) & 7] >> 3) ^ (x[(i + j + 1) & 7] & 0x3fu); }
      if (NL > 0 && (i % (32 / (NL > 0 ? NL : 1))) == 0) { b += lds[(lo + (x[0] & 1023) + i) & 2047]; }
    }
  }
  double s = 0;
  for (int i = 0; i < 32; i++) s += acc[i];
  for (int i = 0; i < 8; i++) s += x[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s + b;
}

int main() {
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  int ncu = p.multiProcessorCount; const int iters = 4000;
  double *out; CK(hipMalloc(&out, sizeof(double) * 256 * ncu * 16));
  auto timeit = [&](auto launch) { hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); launch(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0)); for (int i = 0; i < 5; i++) launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); return (double)ms / 5; };
#define RUN(NI, NL, BPC) { int blocks = ncu * BPC; double ms = timeit([&] { k<NI, NL><<<blocks, 256>>>(out, iters, 1.0, 0.5, 12345u); }); \
  printf("int-ops/mfma=%d (x2 instr) lds-reads/32mfma=%d blocks/CU=%d: %.3f ms  %.2f TFLOP/s\n", NI, NL, BPC, ms, (double)blocks * 4 * iters * 512.0 * 32 / ms * 1e-9); }
  RUN(0, 0, 1) RUN(1, 0, 1) RUN(2, 0, 1) RUN(3, 0, 1) RUN(4, 0, 1) RUN(6, 0, 1)
  RUN(0, 0, 2) RUN(1, 0, 2) RUN(2, 0, 2) RUN(3, 0, 2) RUN(4, 0, 2) RUN(6, 0, 2)
  RUN(0, 4, 1) RUN(0, 8, 1) RUN(0, 16, 1) RUN(0, 32, 1) RUN(1, 8, 1) RUN(1, 8, 2) RUN(2, 16, 2)
  return 0;
}
