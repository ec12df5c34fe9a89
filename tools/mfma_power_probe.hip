// mfma_power_probe.hip -- what does the chip SUSTAIN on a bare stream of matrix-core instructions, with no memory traffic at all?
//   mfma_power_probe <f4|i8|f64> <seconds> [waves_per_simd = 1] [pattern = geno|zero|rand]
// One workgroup of 256 threads (4 waves, one per SIMD) per CU slot; every wave runs 16 independent accumulator chains of
//   f4 : v_mfma_scale_f32_32x32x64_f8f6f4 (FP4 e2m1 operands, unit scales)   -- the instruction of k_crossprod_f4
//   i8 : v_mfma_i32_32x32x32_i8                                               -- k_crossprod_i8 / k_gemm_i8
//   f64: v_mfma_f64_4x4x4_4b_f64                                              -- k_gemm
// with operands held in registers (geno: nibbles / bytes in {0,1,2} like genotypes; rand: random bits; zero).  Launches of ~20 ms are repeated for
// <seconds>; prints the rate of every ~0.5 s window.  Run under tools/power_trace.py cmd to see power and clock beside it: the rate this loop
// holds is the ceiling any real kernel on that instruction can approach on this board (profiles/r06_power_*.txt).
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef int v4i __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef int v16i __attribute__((ext_vector_type(16)));

__device__ __forceinline__ uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
// pattern 0 geno (fields in {0,1,2}), 1 zero, 2 random bits; `field` = bits per value (4: nibbles, 8: bytes)
__device__ __forceinline__ int operand_word(uint32_t seed, int pattern, int field) {
  if (pattern == 1) return 0;
  uint32_t r = mix(seed);
  if (pattern == 2) return (int)r;
  uint32_t w = 0;
  for (int b = 0; b < 32; b += field) { w |= ((r % 3u) << b); r = mix(r + b); }
  return (int)w;
}

template <int KIND>   // 0 f4, 1 i8, 2 f64
__global__ void __launch_bounds__(256) k_probe(int iters, int pattern, float *sink) {
  const uint32_t id = blockIdx.x * 256u + threadIdx.x;
  if (KIND == 0) {
    v16f acc[16];
    for (int t = 0; t < 16; t++) for (int r = 0; r < 16; r++) acc[t][r] = 0.f;
    v8i a[2], b[2];
    for (int q = 0; q < 2; q++) for (int r = 0; r < 8; r++) {
      a[q][r] = r < 4 ? operand_word(id * 16 + q * 4 + r, pattern, 4) : 0;
      b[q][r] = r < 4 ? operand_word(id * 16 + 8 + q * 4 + r, pattern, 4) : 0;
    }
    for (int i = 0; i < iters; i++) {
#pragma unroll
      for (int t = 0; t < 16; t++) acc[t] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[t & 1], b[(t >> 1) & 1], acc[t], 4, 4, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
    }
    float s = 0.f;
    for (int t = 0; t < 16; t++) for (int r = 0; r < 16; r++) s += acc[t][r];
    if (s == 12345.678f) sink[0] = s;
  } else if (KIND == 1) {
    v16i acc[16];
    for (int t = 0; t < 16; t++) for (int r = 0; r < 16; r++) acc[t][r] = 0;
    v4i a[2], b[2];
    for (int q = 0; q < 2; q++) for (int r = 0; r < 4; r++) {
      a[q][r] = operand_word(id * 16 + q * 4 + r, pattern, 8);
      b[q][r] = operand_word(id * 16 + 8 + q * 4 + r, pattern, 8);
    }
    for (int i = 0; i < iters; i++) {
#pragma unroll
      for (int t = 0; t < 16; t++) acc[t] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[t & 1], b[(t >> 1) & 1], acc[t], 0, 0, 0);
    }
    int s = 0;
    for (int t = 0; t < 16; t++) for (int r = 0; r < 16; r++) s += acc[t][r];
    if (s == 123456789) sink[0] = (float)s;
  } else {
    double acc[16];
    for (int t = 0; t < 16; t++) acc[t] = 0.0;
    double a[2], b[2];
    for (int q = 0; q < 2; q++) {
      a[q] = pattern == 1 ? 0.0 : (double)(mix(id * 4 + q) % 3u);
      b[q] = pattern == 1 ? 0.0 : 1.0 + (double)(mix(id * 4 + 2 + q) & 0xffffu) / 65536.0;
    }
    for (int i = 0; i < iters; i++) {
#pragma unroll
      for (int t = 0; t < 16; t++) acc[t] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[t & 1], b[(t >> 1) & 1], acc[t], 0, 0, 0);
    }
    double s = 0.0;
    for (int t = 0; t < 16; t++) s += acc[t];
    if (s == 12345.678) sink[0] = (float)s;
  }
}

int main(int argc, char **argv) {
  if (argc < 3) { fprintf(stderr, "usage: %s <f4|i8|f64> <seconds> [waves_per_simd] [geno|zero|rand]\n", argv[0]); return 2; }
  const std::string kind = argv[1];
  const double seconds = atof(argv[2]);
  const int wps = argc > 3 ? atoi(argv[3]) : 1;
  const std::string pat = argc > 4 ? argv[4] : "geno";
  const int pattern = pat == "zero" ? 1 : pat == "rand" ? 2 : 0;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, 0) != hipSuccess) { fprintf(stderr, "no device\n"); return 1; }
  const int cus = prop.multiProcessorCount, grid = cus * wps;
  float *sink = nullptr;
  (void)hipMalloc(&sink, 4);
  // ops per MFMA: f4 2*32*32*64, i8 2*32*32*32, f64 (4 blocks of 4x4x4) 2*4*4*4*4
  const double ops_per = kind == "f4" ? 2.0 * 32 * 32 * 64 : kind == "i8" ? 2.0 * 32 * 32 * 32 : 2.0 * 4 * 4 * 4 * 4;
  const int iters = kind == "f64" ? 60000 : 6000;       // ~20-40 ms per launch
  auto launch = [&](int it) {
    if (kind == "f4") hipLaunchKernelGGL(k_probe<0>, dim3(grid), dim3(256), 0, 0, it, pattern, sink);
    else if (kind == "i8") hipLaunchKernelGGL(k_probe<1>, dim3(grid), dim3(256), 0, 0, it, pattern, sink);
    else hipLaunchKernelGGL(k_probe<2>, dim3(grid), dim3(256), 0, 0, it, pattern, sink);
  };
  launch(10);
  (void)hipDeviceSynchronize();
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const auto t_start = std::chrono::steady_clock::now();
  double win_ms = 0.0; long win_launches = 0; double all_ms = 0.0; long all_launches = 0;
  printf("# mfma_power_probe %s, %d CUs x %d wave(s) per SIMD, operands %s, %d MFMAs x 16 chains per wave per launch\n", kind.c_str(), cus, wps, pat.c_str(), iters);
  for (;;) {
    (void)hipEventRecord(e0);
    launch(iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    win_ms += ms; win_launches++; all_ms += ms; all_launches++;
    const double t = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count();
    if (win_ms >= 500.0 || t >= seconds) {
      const double ops = (double)win_launches * grid * 4.0 * iters * 16.0 * ops_per;
      printf("t=%5.1fs  %.3f %s  (%.2f ms per launch)\n", t, ops / (win_ms * 1e-3) * (kind == "f64" ? 1e-12 : 1e-15), kind == "f64" ? "TFLOP/s" : "Pop/s", win_ms / win_launches);
      fflush(stdout);
      win_ms = 0.0; win_launches = 0;
    }
    if (t >= seconds) break;
  }
  const double ops = (double)all_launches * grid * 4.0 * iters * 16.0 * ops_per;
  printf("mean over %.1f s: %.3f %s\n", all_ms * 1e-3, ops / (all_ms * 1e-3) * (kind == "f64" ? 1e-12 : 1e-15), kind == "f64" ? "TFLOP/s" : "Pop/s");
  return 0;
}
