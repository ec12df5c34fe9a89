// Microbenchmark + layout probe for gfx950 fp64 matrix/vector pipes.
// Build: hipcc --offload-arch=gfx950 -O3 tools/mfma_f64_probe.hip -o tools/mfma_f64_probe
// Measures: v_mfma_f64_16x16x4_f64 issue rate (1..8 independent accumulators, 1/2 waves per SIMD),
// v_mfma_f64_4x4x4_4b_f64, v_fma_f64, and MFMA+VALU-FMA co-issue. Also checks the operand lane map.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <int NACC>
__global__ void __launch_bounds__(256) k_mfma16(double *out, int iters, double a0, double b0) {
  d4 acc[NACC];
  for (int i = 0; i < NACC; i++) acc[i] = d4{0, 0, 0, 0};
  double a = a0 + threadIdx.x * 1e-9, b = b0;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < NACC; i++) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  double s = 0;
  for (int i = 0; i < NACC; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC>
__global__ void __launch_bounds__(256) k_mfma4(double *out, int iters, double a0, double b0) {
  double acc[NACC];
  for (int i = 0; i < NACC; i++) acc[i] = 0;
  double a = a0 + threadIdx.x * 1e-9, b = b0;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < NACC; i++) acc[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[i], 0, 0, 0);
  }
  double s = 0;
  for (int i = 0; i < NACC; i++) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC>
__global__ void __launch_bounds__(256) k_fma(double *out, int iters, double a0, double b0) {
  double acc[NACC];
  for (int i = 0; i < NACC; i++) acc[i] = i;
  double a = a0 + threadIdx.x * 1e-9, b = b0;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < NACC; i++) acc[i] = __builtin_fma(a, acc[i], b);
  }
  double s = 0;
  for (int i = 0; i < NACC; i++) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// NM MFMAs + NF VALU FMAs per iteration, independent chains
template <int NM, int NF>
__global__ void __launch_bounds__(256) k_mix(double *out, int iters, double a0, double b0) {
  d4 acc[NM];
  double f[NF > 0 ? NF : 1];
  for (int i = 0; i < NM; i++) acc[i] = d4{0, 0, 0, 0};
  for (int i = 0; i < NF; i++) f[i] = i;
  double a = a0 + threadIdx.x * 1e-9, b = b0;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < (NM > NF ? NM : NF); i++) {
      if (i < NM) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
      if (i < NF) f[i] = __builtin_fma(a, f[i], b);
    }
  }
  double s = 0;
  for (int i = 0; i < NM; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  for (int i = 0; i < NF; i++) s += f[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// layout probe: D = A(16x4) * B(4x16), A[i][k] = 1 + i + 100k... exact integers
__global__ void k_layout(const double *A, const double *B, double *D) {
  int l = threadIdx.x;
  double a = A[(l & 15) * 4 + (l >> 4)];   // A[i=l&15][k=l>>4]
  double b = B[(l >> 4) * 16 + (l & 15)];  // B[k=l>>4][j=l&15]
  d4 c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
  for (int r = 0; r < 4; r++) D[((l >> 4) + 4 * r) * 16 + (l & 15)] = c[r];  // row=(l>>4)+4r, col=l&15
}

template <typename F>
double time_kernel(F launch, int reps) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  launch();
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int i = 0; i < reps; i++) launch();
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  return ms / reps;
}

int main() {
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  printf("device %s CUs %d clock %d kHz\n", p.name, p.multiProcessorCount, p.clockRate);
  int ncu = p.multiProcessorCount;
  double *out; CK(hipMalloc(&out, sizeof(double) * 256 * ncu * 16));
  const int iters = 20000;
  // layout
  {
    std::vector<double> A(64), B(64), D(256), R(256, 0.0);
    for (int i = 0; i < 16; i++) for (int k = 0; k < 4; k++) A[i * 4 + k] = 1 + i + 3 * k;
    for (int k = 0; k < 4; k++) for (int j = 0; j < 16; j++) B[k * 16 + j] = 2 + 5 * k + 7 * j + (j * j);
    for (int i = 0; i < 16; i++) for (int j = 0; j < 16; j++) for (int k = 0; k < 4; k++) R[i * 16 + j] += A[i * 4 + k] * B[k * 16 + j];
    double *dA, *dB, *dD; CK(hipMalloc(&dA, 512)); CK(hipMalloc(&dB, 512)); CK(hipMalloc(&dD, 2048));
    CK(hipMemcpy(dA, A.data(), 512, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), 512, hipMemcpyHostToDevice));
    k_layout<<<1, 64>>>(dA, dB, dD);
    CK(hipMemcpy(D.data(), dD, 2048, hipMemcpyDeviceToHost));
    int bad = 0; for (int i = 0; i < 256; i++) bad += (D[i] != R[i]);
    printf("layout probe: %d mismatches of 256 (A[i=l&15][k=l>>4], B[k=l>>4][j=l&15], D row=(l>>4)+4r col=l&15)\n", bad);
  }
  auto report = [&](const char *name, double ms, double flop_per_thread_iter_wave, int blocks, int nwaves_per_block, double inst_per_iter) {
    double waves = (double)blocks * nwaves_per_block;
    double flops = waves * iters * flop_per_thread_iter_wave;
    double cyc_per_inst = ms * 1e-3 * 2.4e9 / (iters * inst_per_iter) ;  // per wave at nominal 2.4 GHz
    printf("%-40s %8.3f ms  %8.2f TFLOP/s   ~%6.1f cyc/inst/wave@2.4GHz\n", name, ms, flops / ms * 1e-9, cyc_per_inst);
  };
#define RUN_MFMA16(N, BPC) { int blocks = ncu * BPC; double ms = time_kernel([&] { k_mfma16<N><<<blocks, 256>>>(out, iters, 1.0, 0.5); }, 5); \
    char nm[64]; snprintf(nm, 64, "mfma16x16x4 acc=%d blocks/CU=%d", N, BPC); report(nm, ms, 2048.0 * N, blocks, 4, N); }
  RUN_MFMA16(1, 1) RUN_MFMA16(2, 1) RUN_MFMA16(4, 1) RUN_MFMA16(8, 1) RUN_MFMA16(1, 2) RUN_MFMA16(4, 2) RUN_MFMA16(8, 2) RUN_MFMA16(4, 4)
#define RUN_MFMA4(N, BPC) { int blocks = ncu * BPC; double ms = time_kernel([&] { k_mfma4<N><<<blocks, 256>>>(out, iters, 1.0, 0.5); }, 5); \
    char nm[64]; snprintf(nm, 64, "mfma4x4x4_4b acc=%d blocks/CU=%d", N, BPC); report(nm, ms, 512.0 * N, blocks, 4, N); }
  RUN_MFMA4(1, 1) RUN_MFMA4(4, 1) RUN_MFMA4(8, 1) RUN_MFMA4(8, 2)
#define RUN_FMA(N, BPC) { int blocks = ncu * BPC; double ms = time_kernel([&] { k_fma<N><<<blocks, 256>>>(out, iters, 1.0, 0.5); }, 5); \
    char nm[64]; snprintf(nm, 64, "v_fma_f64 chains=%d blocks/CU=%d", N, BPC); report(nm, ms, 128.0 * N, blocks, 4, N); }
  RUN_FMA(8, 1) RUN_FMA(16, 1) RUN_FMA(16, 2) RUN_FMA(16, 4)
#define RUN_MIX(NM, NF, BPC) { int blocks = ncu * BPC; double ms = time_kernel([&] { k_mix<NM, NF><<<blocks, 256>>>(out, iters, 1.0, 0.5); }, 5); \
    char nm[64]; snprintf(nm, 64, "mix mfma=%d fma=%d blocks/CU=%d", NM, NF, BPC); report(nm, ms, 2048.0 * NM + 128.0 * NF, blocks, 4, NM); }
  RUN_MIX(4, 0, 2) RUN_MIX(4, 4, 2) RUN_MIX(4, 16, 2) RUN_MIX(4, 32, 2) RUN_MIX(4, 64, 2) RUN_MIX(4, 64, 1)
  CK(hipFree(out));
  return 0;
}
