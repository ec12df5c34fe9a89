// hbm_pattern_probe.hip -- what does the ACCESS PATTERN of the two HBM-bound int8 kernels cost, with nothing but the LDS-DMA stream left?
//   plain  (k_gemm_i8, 'T'):      one workgroup per row block of the tiled packed matrix, 4 per CU; it streams its row block's tiles front to back
//                                 (8 KiB per stage, contiguous from stage to stage), 2 stages in flight
//   tn     (k_gemm_i8_tn, 'N'):   persistent workgroups, 2 per CU, over items = (strip of 2 slabs, piece of the row blocks), piece-major as in the kernel; a stage = the two adjacent
//                                 tiles (row block rb, slabs 2 strip, 2 strip + 1) = 16 KiB, the next stage is nslabs x 8 KiB further on; 3 stages in flight
// Same bytes (the whole matrix once), same `nt` LDS-DMA units of 1 KiB, no arithmetic, no barrier.  The ratio of the two rates is what the transposed-operand kernel pays for reading the
// SNP-major copy across its rows.   usage: hbm_pattern_probe [snps = 250000] [indiv = 100000] [reps = 5] [only = -1: all seven patterns; 0 .. 6: that one alone -- with thousands of reps: a steady load for tools/power_trace.py hbm_*]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
using lptr_t = __attribute__((address_space(3))) void *;
__device__ __forceinline__ void dma16_nt(const void *sbase, uint32_t voff, uint32_t lds_addr) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %0 nt" ::"s"(sbase), "v"(voff), "s"(lds_addr) : "memory", "m0");
}
constexpr int kTile = 8192;

// plain: block = row block rb; wave w issues units w, w + 4 of every tile; NB buffers of 8 KiB
template <int NB>
__global__ void __launch_bounds__(256) k_plain(const char *__restrict__ G, long nslabs, int rowblocks) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint32_t lds0 = (uint32_t)(size_t)(lptr_t)smem;
  for (int rb = blockIdx.x; rb < rowblocks; rb += gridDim.x) {
    const char *src = G + (size_t)rb * nslabs * kTile;
    for (long s = 0; s < nslabs; s++) {
#pragma unroll
      for (int i = 0; i < 2; i++) dma16_nt(src + s * kTile + (wave + 4 * i) * 1024, lane * 16, lds0 + (uint32_t)(s % NB) * kTile + (wave + 4 * i) * 1024);
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NB - 1) * 2) : "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
}

// tn: items (strip, piece), piece-major; per stage the wave's 4 units (K-steps 2 w, 2 w + 1 of both slabs); NB buffers of 16 KiB, NB - 1 stages in flight.
// DIG: + the wave's two digit fragments of the stage (2 x 1 KiB straight into registers from an L2-resident array, as k_gemm_i8_tn loads them)
typedef int v4i __attribute__((ext_vector_type(4)));
template <int NB, bool DIG>
__global__ void __launch_bounds__(256) k_tn(const char *__restrict__ G, long nslabs, int rowblocks, int strips, int pieces, const char *__restrict__ Dg, int *__restrict__ sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint32_t lds0 = (uint32_t)(size_t)(lptr_t)smem;
  const int nitems = strips * pieces;
  for (int item = blockIdx.x; item < nitems; item += gridDim.x) {
    const int piece = item / strips, strip = item - piece * strips;
    const int st0 = (int)((long)piece * rowblocks / pieces), st1 = (int)((long)(piece + 1) * rowblocks / pieces);
    for (int s = st0; s < st1; s++) {
#pragma unroll
      for (int i = 0; i < 4; i++) {
        const int u = (i >> 1) * 8 + 2 * wave + (i & 1);
        long sl = 2L * strip + (u >> 3);
        if (sl >= nslabs) sl = nslabs - 1;
        dma16_nt(G + ((size_t)s * nslabs + sl) * kTile + (u & 7) * 1024, lane * 16, lds0 + (uint32_t)((s - st0) % NB) * (2 * kTile) + u * 1024);
      }
      if (DIG) {
        v4i d0, d1;
        const char *dsrc = Dg + ((size_t)s * 8 + 2 * wave) * 1024;
        asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(d0) : "v"(lane * 16), "s"(dsrc) : "memory");
        asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(d1) : "v"(lane * 16), "s"(dsrc + 1024) : "memory");
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NB - 1) * 6) : "memory");
        asm volatile("" :: "v"(d0), "v"(d1));
      } else
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NB - 1) * 4) : "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  if (sink && blockIdx.x == 0x7fffffff) sink[0] = 1;
}
// plain + the stage's digit fragments through LDS (4 more units of 1 KiB per stage and workgroup, as k_gemm_i8 stages them)
template <int NB>
__global__ void __launch_bounds__(256) k_plain_dig(const char *__restrict__ G, long nslabs, int rowblocks, const char *__restrict__ Dg) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint32_t lds0 = (uint32_t)(size_t)(lptr_t)smem;
  for (int rb = blockIdx.x; rb < rowblocks; rb += gridDim.x) {
    const char *src = G + (size_t)rb * nslabs * kTile;
    for (long s = 0; s < nslabs; s++) {
      const uint32_t base = lds0 + (uint32_t)(s % NB) * (kTile + 4096);
#pragma unroll
      for (int i = 0; i < 2; i++) dma16_nt(src + s * kTile + (wave + 4 * i) * 1024, lane * 16, base + (wave + 4 * i) * 1024);
      asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %0" ::"s"(Dg + ((size_t)s * 4 + wave) * 1024), "v"(lane * 16), "s"(base + kTile + wave * 1024) : "memory", "m0");
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NB - 1) * 3) : "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
}

int main(int argc, char **argv) {
  const long snps = argc > 1 ? atol(argv[1]) : 250000, indiv = argc > 2 ? atol(argv[2]) : 100000;
  const int reps = argc > 3 ? atoi(argv[3]) : 5, only = argc > 4 ? atoi(argv[4]) : -1;
  const long nslabs = (indiv + 127) / 128;
  const int rowblocks = (int)((snps + 255) / 256);
  const size_t bytes = (size_t)rowblocks * nslabs * kTile;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, 0) != hipSuccess) { fprintf(stderr, "no device\n"); return 1; }
  const int cus = prop.multiProcessorCount;
  char *G = nullptr;
  if (hipMalloc(&G, bytes) != hipSuccess) { fprintf(stderr, "hipMalloc of %zu bytes failed\n", bytes); return 1; }
  (void)hipMemset(G, 1, bytes);
  const int strips = (int)((nslabs + 1) / 2);
  // pieces so that the items fill whole rounds of the 2 x CUs slots (the kernel's planner does the same): smallest c with strips * c >= a multiple of the slots
  int pieces = 1;
  while ((long)strips * pieces < 4L * 2 * cus) pieces++;
  printf("# hbm_pattern_probe: %ld SNPs x %ld individuals = %d row blocks x %ld slabs of 8 KiB tiles = %.2f GB; %d CUs; tn: %d strips x %d pieces\n", snps, indiv, rowblocks, nslabs, bytes / 1e9, cus, strips, pieces);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  int idx = 0;
  auto time_it = [&](const char *name, auto launch) {
    if (only >= 0 && idx++ != only) return;
    launch();
    (void)hipDeviceSynchronize();
    float best = 1e30f, sum = 0.f;
    for (int r = 0; r < reps; r++) {
      (void)hipEventRecord(e0);
      launch();
      (void)hipEventRecord(e1);
      (void)hipEventSynchronize(e1);
      float ms = 0.f;
      (void)hipEventElapsedTime(&ms, e0, e1);
      best = std::min(best, ms); sum += ms;
    }
    printf("%-44s best %.3f ms = %.2f TB/s, mean %.3f ms = %.2f TB/s\n", name, best, bytes / (best * 1e-3) * 1e-12, sum / reps, bytes / (sum / reps * 1e-3) * 1e-12);
  };
  char *Dg = nullptr;                                        // the digit fragments: 8 KiB per row block ('N': K runs over the SNP rows), 4 KiB per slab ('T')
  const size_t dbytes = std::max((size_t)rowblocks * 8192, (size_t)nslabs * 4096);
  (void)hipMalloc(&Dg, dbytes);
  (void)hipMemset(Dg, 2, dbytes);
  (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_tn<4, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 2 * kTile);
  (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_tn<4, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 2 * kTile);
  (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_tn<3, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 2 * kTile);
  time_it("plain: row-block streams, 4 WG/CU, 3 buffers", [&] { hipLaunchKernelGGL(k_plain<3>, dim3(std::min(rowblocks, 4 * cus)), dim3(256), 3 * kTile, 0, G, nslabs, rowblocks); });
  time_it("plain: row-block streams, 4 WG/CU, 4 buffers", [&] { hipLaunchKernelGGL(k_plain<4>, dim3(std::min(rowblocks, 4 * cus)), dim3(256), 4 * kTile, 0, G, nslabs, rowblocks); });
  time_it("plain + digits through LDS (4 KiB per stage)", [&] { hipLaunchKernelGGL(k_plain_dig<3>, dim3(std::min(rowblocks, 4 * cus)), dim3(256), 3 * (kTile + 4096), 0, G, nslabs, rowblocks, Dg); });
  time_it("tn: strips across the rows, 2 WG/CU, 4 buffers", [&] { hipLaunchKernelGGL((k_tn<4, false>), dim3(2 * cus), dim3(256), 4 * 2 * kTile, 0, G, nslabs, rowblocks, strips, pieces, Dg, (int *)nullptr); });
  time_it("tn + digits into registers (8 KiB per stage)", [&] { hipLaunchKernelGGL((k_tn<4, true>), dim3(2 * cus), dim3(256), 4 * 2 * kTile, 0, G, nslabs, rowblocks, strips, pieces, Dg, (int *)nullptr); });
  time_it("tn: strips across the rows, 3 WG/CU, 3 buffers", [&] { hipLaunchKernelGGL((k_tn<3, false>), dim3(3 * cus), dim3(256), 3 * 2 * kTile, 0, G, nslabs, rowblocks, strips, pieces, Dg, (int *)nullptr); });
  time_it("plain again (order check)", [&] { hipLaunchKernelGGL(k_plain<3>, dim3(std::min(rowblocks, 4 * cus)), dim3(256), 3 * kTile, 0, G, nslabs, rowblocks); });
  return 0;
}
