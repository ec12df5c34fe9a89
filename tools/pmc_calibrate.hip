// pmc_calibrate.hip -- known-size streams for calibrating rocprofv3's FETCH_SIZE on gfx950 in the access patterns this library uses.
//   k_stream_dma   : global_load_lds_dwordx4, wave-uniform base + lane*16 (1 KiB contiguous per wave instruction) -- k_gemm's operand path
//   k_stream_dma32 : global_load_lds_dwordx4, lane*32 stride (16 B of every 32-B row piece per instruction, two instructions cover a
//                    2 KiB run) -- the row path of the small-n kernel
//   k_stream_vec   : plain global_load_dwordx4, 16 B per lane, coalesced -- the pattern the guide's x2 correction was measured on
// Every kernel reads `bytes` exactly once.  Build: hipcc --offload-arch=gfx950 -O3 -o tools/pmc_calibrate tools/pmc_calibrate.hip
// Run under `rocprofv3 --pmc FETCH_SIZE --kernel-trace -- tools/pmc_calibrate <GiB>`; tools/pmc_calibrate.py parses the csv.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>

using lptr_t = __attribute__((address_space(3))) void *;
__device__ __forceinline__ void dma16_s(const void *sbase, uint32_t voff, uint32_t lds_addr) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %0" ::"s"(sbase), "v"(voff), "s"(lds_addr) : "memory", "m0");
}

__device__ __forceinline__ void dma16_nt(const void *sbase, uint32_t voff, uint32_t lds_addr) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %0 nt" ::"s"(sbase), "v"(voff), "s"(lds_addr) : "memory", "m0");
}

// each workgroup (256 threads) streams a contiguous 64 KiB piece per iteration: 16 units of 1 KiB per wave
template <int STRIDE32>
__global__ void __launch_bounds__(256) k_stream_dma(const char *__restrict__ src, size_t pieces) {
  __shared__ __attribute__((aligned(16))) char buf[65536];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint32_t lds0 = (uint32_t)(size_t)(lptr_t)buf;
  for (size_t p = blockIdx.x; p < pieces; p += gridDim.x) {
    const char *base = src + p * 65536 + (size_t)wave * 16384;
#pragma unroll
    for (int u = 0; u < 16; u++) {
      if (STRIDE32 == 1) dma16_s(base + (u >> 1) * 2048 + (u & 1) * 16, lane * 32, lds0 + wave * 16384 + u * 1024);
      else if (STRIDE32 == 2) dma16_nt(base + u * 1024, lane * 16, lds0 + wave * 16384 + u * 1024);   // round 3: the non-temporal hint (k_gemm_i8's packed stream)
      else dma16_s(base + u * 1024, lane * 16, lds0 + wave * 16384 + u * 1024);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
}

__global__ void __launch_bounds__(256) k_stream_vec(const uint4 *__restrict__ src, size_t n16, unsigned *__restrict__ sink) {
  unsigned acc = 0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) {
    const uint4 v = src[i];
    acc ^= v.x ^ v.y ^ v.z ^ v.w;
  }
  if (acc == 0x12345678u) sink[0] = acc;   // keeps the loads alive
}

int main(int argc, char **argv) {
  const double gib = argc > 1 ? atof(argv[1]) : 8.0;
  const size_t bytes = ((size_t)(gib * (1ull << 30)) >> 16) << 16;
  char *d = nullptr;
  unsigned *sink = nullptr;
  if (hipMalloc((void **)&d, bytes) != hipSuccess || hipMalloc((void **)&sink, 4) != hipSuccess) { printf("hipMalloc failed\n"); return 1; }
  hipMemset(d, 1, bytes);
  hipDeviceSynchronize();
  const size_t pieces = bytes >> 16;
  for (int rep = 0; rep < 2; rep++) {
    hipLaunchKernelGGL(k_stream_dma<0>, dim3(2048), dim3(256), 0, 0, d, pieces);
    hipLaunchKernelGGL(k_stream_dma<1>, dim3(2048), dim3(256), 0, 0, d, pieces);
    hipLaunchKernelGGL(k_stream_vec, dim3(4096), dim3(256), 0, 0, (const uint4 *)d, bytes / 16, sink);
    hipLaunchKernelGGL(k_stream_dma<2>, dim3(2048), dim3(256), 0, 0, d, pieces);
  }
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  float ms[3];
  hipEventRecord(e0); hipLaunchKernelGGL(k_stream_dma<0>, dim3(2048), dim3(256), 0, 0, d, pieces); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms[0], e0, e1);
  hipEventRecord(e0); hipLaunchKernelGGL(k_stream_dma<1>, dim3(2048), dim3(256), 0, 0, d, pieces); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms[1], e0, e1);
  hipEventRecord(e0); hipLaunchKernelGGL(k_stream_vec, dim3(4096), dim3(256), 0, 0, (const uint4 *)d, bytes / 16, sink); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms[2], e0, e1);
  if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 1; }
  printf("CALIB bytes %zu  dma_linear %.3f ms (%.2f TB/s)  dma_stride32 %.3f ms (%.2f TB/s)  vec16 %.3f ms (%.2f TB/s)\n", bytes, ms[0], bytes / ms[0] * 1e-9,
         ms[1], bytes / ms[1] * 1e-9, ms[2], bytes / ms[2] * 1e-9);
  return 0;
}
