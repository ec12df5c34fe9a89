// hbm_read_probe.hip -- how fast can ONE MI355X read a large buffer exactly once?  (ceiling for the HBM-bound kernels: k_gemm_i8 at n <= 6, k_lut)
//   vec<U, NT>   : plain global_load_dwordx4, U independent 16-byte loads per lane in flight, grid-stride over 4 KiB * U pieces; NT = nontemporal
//   dma<DEPTH>   : global_load_lds_dwordx4 (lane-linear 1-KiB units), every wave keeps DEPTH units of 1 KiB in flight in its own LDS ring
//                  (no consumer, no barrier: the pure issue rate of the operand path of k_gemm / k_gemm_i8)
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/hbm_read_probe tools/hbm_read_probe.hip ; run: tools/hbm_read_probe [GiB]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
using lptr_t = __attribute__((address_space(3))) void *;
__device__ __forceinline__ void dma16_s(const void *sbase, uint32_t voff, uint32_t lds_addr) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %0" ::"s"(sbase), "v"(voff), "s"(lds_addr) : "memory", "m0");
}
template <int U, bool NT>
__global__ void __launch_bounds__(256) k_vec(const uint4 *__restrict__ src, size_t npieces, unsigned *__restrict__ sink) {
  unsigned acc = 0;
  for (size_t p = blockIdx.x; p < npieces; p += gridDim.x) {
    const uint4 *b = src + p * (256 * U) + threadIdx.x;
    uint4 v[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
      if (NT) { typedef unsigned v4u __attribute__((ext_vector_type(4))); const v4u t = __builtin_nontemporal_load(reinterpret_cast<const v4u *>(b + u * 256)); v[u] = make_uint4(t.x, t.y, t.z, t.w); }
      else v[u] = b[u * 256];
    }
#pragma unroll
    for (int u = 0; u < U; u++) acc ^= v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
  }
  if (acc == 0x12345678u) sink[0] = acc;
}
template <int POL> __device__ __forceinline__ void dma16_p(const void *sbase, uint32_t voff, uint32_t lds_addr) {
  if (POL == 1) asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %0 nt" ::"s"(sbase), "v"(voff), "s"(lds_addr) : "memory", "m0");
  else if (POL == 2) asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %0 sc1" ::"s"(sbase), "v"(voff), "s"(lds_addr) : "memory", "m0");
  else if (POL == 3) asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %0 sc0 sc1" ::"s"(sbase), "v"(voff), "s"(lds_addr) : "memory", "m0");
  else if (POL == 4) asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %0 sc0 sc1 nt" ::"s"(sbase), "v"(voff), "s"(lds_addr) : "memory", "m0");
  else if (POL == 5) asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %0 sc1 nt" ::"s"(sbase), "v"(voff), "s"(lds_addr) : "memory", "m0");
  else dma16_s(sbase, voff, lds_addr);
}
template <int DEPTH, int POL = 0>
__global__ void __launch_bounds__(256) k_dma(const char *__restrict__ src, size_t nunits_per_wave_total, size_t total_waves) {
  extern __shared__ __attribute__((aligned(16))) char buf[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint32_t lds0 = (uint32_t)(size_t)(lptr_t)buf + wave * DEPTH * 1024;
  const size_t gw = (size_t)blockIdx.x * 4 + wave;          // global wave id; units are dealt round-robin in runs of 16 KiB per wave
  // wave w reads 16-KiB runs number w, w + total_waves, ...
  const size_t runs = nunits_per_wave_total;                 // runs of 16 units per wave
  int slot = 0;
  for (size_t r = 0; r < runs; r++) {
    const char *base = src + ((r * total_waves + gw) << 14);
#pragma unroll
    for (int u = 0; u < 16; u++) {
      dma16_p<POL>(base + u * 1024, lane * 16, lds0 + slot * 1024);
      slot = slot + 1 == DEPTH ? 0 : slot + 1;
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DEPTH - 1) : "memory");
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
static float timeit(void (*launch)(), int reps) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  launch(); hipDeviceSynchronize();
  float best = 1e30f;
  for (int i = 0; i < reps; i++) { hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms; }
  return best;
}
static char *g_d; static unsigned *g_sink; static size_t g_bytes; static int g_grid;
template <int U, bool NT> static void l_vec() { hipLaunchKernelGGL((k_vec<U, NT>), dim3(g_grid), dim3(256), 0, 0, (const uint4 *)g_d, g_bytes / (4096 * U), g_sink); }
template <int DEPTH, int POL = 0> static void l_dma() {
  const size_t total_waves = (size_t)g_grid * 4;
  hipLaunchKernelGGL((k_dma<DEPTH, POL>), dim3(g_grid), dim3(256), 4 * DEPTH * 1024, 0, g_d, g_bytes / (total_waves << 14), total_waves);
}
int main(int argc, char **argv) {
  const double gib = argc > 1 ? atof(argv[1]) : 8.0;
  g_bytes = ((size_t)(gib * (1ull << 30)) >> 26) << 26;
  if (hipMalloc((void **)&g_d, g_bytes) != hipSuccess || hipMalloc((void **)&g_sink, 4) != hipSuccess) { printf("hipMalloc failed\n"); return 1; }
  hipMemset(g_d, 1, g_bytes); hipDeviceSynchronize();
  for (int grid : {1024, 2048, 4096, 8192}) {
    g_grid = grid;
    printf("HBMPROBE grid %5d  vec U=2 %.2f  U=4 %.2f  U=8 %.2f  U=8nt %.2f  U=16 %.2f TB/s", grid, g_bytes / timeit(l_vec<2, false>, 3) * 1e-9, g_bytes / timeit(l_vec<4, false>, 3) * 1e-9,
           g_bytes / timeit(l_vec<8, false>, 3) * 1e-9, g_bytes / timeit(l_vec<8, true>, 3) * 1e-9, g_bytes / timeit(l_vec<16, false>, 3) * 1e-9);
    printf("\n");
  }
  for (int grid : {256, 512, 1024, 2048}) {
    g_grid = grid;
    printf("HBMPROBE grid %5d  dma depth 4 %.2f  8 %.2f  16 %.2f  32 %.2f TB/s\n", grid, g_bytes / timeit(l_dma<4>, 3) * 1e-9, g_bytes / timeit(l_dma<8>, 3) * 1e-9,
           g_bytes / timeit(l_dma<16>, 3) * 1e-9, g_bytes / timeit(l_dma<32>, 3) * 1e-9);
  }
  for (int grid : {512, 2048}) {
    g_grid = grid;
    printf("HBMPROBE grid %5d  dma depth 16: default %.2f  nt %.2f  sc1 %.2f  sc0 sc1 %.2f  sc0 sc1 nt %.2f  sc1 nt %.2f TB/s\n", grid, g_bytes / timeit(l_dma<16, 0>, 3) * 1e-9,
           g_bytes / timeit(l_dma<16, 1>, 3) * 1e-9, g_bytes / timeit(l_dma<16, 2>, 3) * 1e-9, g_bytes / timeit(l_dma<16, 3>, 3) * 1e-9, g_bytes / timeit(l_dma<16, 4>, 3) * 1e-9,
           g_bytes / timeit(l_dma<16, 5>, 3) * 1e-9);
  }
  if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 1; }
  return 0;
}
