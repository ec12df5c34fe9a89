// mall_replay_probe.hip -- can a SECOND read of the packed matrix be served from the 256 MiB Infinity Cache if it follows the first read closely?
// (the fused CG step G v = Zc (Zc^T v) needs every packed byte twice: once for t = Zc^T v and, after t of the row block is complete, once for
// y += Zc t; DESIGN.md "single-pass G v").  The probe reads a buffer of `GiB` in chunks of `chunk_MB`; every chunk is read twice, the second time
// `lag` chunks later (order P1(c0) .. P1(c_lag-1), then alternating P2(c), P1(c + lag)).  No arithmetic, no dependencies: the pure memory side.
// LDS-DMA streams as in k_gemm_i8 (4 waves per workgroup, DEPTH 1-KiB units in flight per wave).  Reported: TB/s over 2 x the buffer.
//   lag = number of chunks (>= nchunks): two plain passes (baseline).
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/mall_replay_probe tools/mall_replay_probe.hip ; run: tools/mall_replay_probe [GiB]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
using lptr_t = __attribute__((address_space(3))) void *;
template <int POL> __device__ __forceinline__ void dma16_p(const void *sbase, uint32_t voff, uint32_t lds_addr) {
  if (POL == 1) asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %0 nt" ::"s"(sbase), "v"(voff), "s"(lds_addr) : "memory", "m0");
  else asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %0" ::"s"(sbase), "v"(voff), "s"(lds_addr) : "memory", "m0");
}
// items: phase q = 0 .. 2 nchunks - 1 in the interleaved order; inside a phase the 16-KiB runs of the chunk are dealt to the waves round-robin
template <int DEPTH, int POL1, int POL2>
__global__ void __launch_bounds__(256) k_replay(const char *__restrict__ src, long nchunks, long runs_per_chunk, long lag, long total_waves, long shift2) {
  extern __shared__ __attribute__((aligned(16))) char buf[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint32_t lds0 = (uint32_t)(size_t)(lptr_t)buf + wave * DEPTH * 1024;
  const long gw = (long)blockIdx.x * 4 + wave;
  int slot = 0;
  for (long q = 0; q < 2 * nchunks; q++) {
    // schedule: the first `lag` phases are first reads; afterwards second reads and first reads alternate until the first reads run out
    long c; bool second;
    if (q < lag) { c = q; second = false; }
    else {
      const long r = q - lag;                    // r = 0: P2(c0), 1: P1(c_lag), 2: P2(c1), ...
      const long firsts_left = nchunks - lag;    // first reads still to do after the prologue
      if (r < 2 * firsts_left) { second = (r & 1) == 0; c = second ? r / 2 : lag + r / 2; }
      else { second = true; c = firsts_left + (r - 2 * firsts_left); }
    }
    const char *cb = src + ((size_t)c * runs_per_chunk << 14);
    // shift2: the SECOND read of a run is made by another workgroup (shift2 = 4 waves: the next block, i.e. normally the next XCD): hits then come
    // from the memory-side Infinity Cache, not from the reading XCD's own L2
    for (long r = second ? (gw + shift2) % total_waves : gw; r < runs_per_chunk; r += total_waves) {
      const char *base = cb + ((size_t)r << 14);
#pragma unroll
      for (int u = 0; u < 16; u++) {
        if (second) dma16_p<POL2>(base + u * 1024, lane * 16, lds0 + slot * 1024);
        else dma16_p<POL1>(base + u * 1024, lane * 16, lds0 + slot * 1024);
        slot = slot + 1 == DEPTH ? 0 : slot + 1;
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DEPTH - 1) : "memory");
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
static char *g_d; static size_t g_bytes; static int g_grid; static long g_nchunks, g_rpc, g_lag, g_shift;
template <int POL1, int POL2> static void l_replay() {
  hipLaunchKernelGGL((k_replay<16, POL1, POL2>), dim3(g_grid), dim3(256), 4 * 16 * 1024, 0, g_d, g_nchunks, g_rpc, g_lag, (long)g_grid * 4, g_shift);
}
static float timeit(void (*launch)(), int reps) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  launch(); hipDeviceSynchronize();
  float best = 1e30f;
  for (int i = 0; i < reps; i++) { hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms; }
  return best;
}
int main(int argc, char **argv) {
  const double gib = argc > 1 ? atof(argv[1]) : 6.0;
  g_bytes = ((size_t)(gib * (1ull << 30)) >> 26) << 26;
  if (hipMalloc((void **)&g_d, g_bytes) != hipSuccess) { printf("hipMalloc failed\n"); return 1; }
  hipMemset(g_d, 1, g_bytes); hipDeviceSynchronize();
  for (int pass = 0; pass < 2; pass++)
  for (int grid : {256, 512, 1024}) {
    g_grid = grid;
    g_shift = pass ? 4 : 0;
    for (int chunk_mb : {16, 32, 64, 128}) {
      g_rpc = ((long)chunk_mb << 20) >> 14;
      g_nchunks = (long)(g_bytes / ((size_t)chunk_mb << 20));
      const double total = 2.0 * (double)g_nchunks * (double)((size_t)chunk_mb << 20);
      for (long lag : {1L, 2L, 3L, 4L, g_nchunks}) {
        if (lag > g_nchunks) continue;
        g_lag = lag;
        const float a = timeit(l_replay<0, 0>, 3), b = timeit(l_replay<0, 1>, 3), c = timeit(l_replay<1, 1>, 3), d = timeit(l_replay<1, 0>, 3);
        printf("MALLPROBE %s grid %4d chunk %3d MB lag %4ld%s: 2 x %.2f GB in  default/default %.3f ms = %.2f TB/s | default/nt %.3f ms = %.2f | nt/nt %.3f ms = %.2f | nt/default %.3f ms = %.2f\n", pass ? "second read by the NEXT workgroup (other XCD)" : "second read by the same wave", grid, chunk_mb, lag,
               lag == g_nchunks ? " (two plain passes)" : "", total / 2e9, a, total / a * 1e-9, b, total / b * 1e-9, c, total / c * 1e-9, d, total / d * 1e-9);
      }
    }
  }
  if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 1; }
  return 0;
}
