// What does it cost to feed the genotype operand of v_mfma_f64_4x4x4_4b_f64 on gfx950, per K-step of 64 MFMAs (8 A x 8 B fragments, like
// k_gemm<8,8>) -- or 16 A x CC B fragments for the narrow tiles?  All operands come from LDS one K-step ahead (two register sets), as in
// a software-pipelined kernel.  Variants:
//   Q0  B fragments by ds_read_b64, A operands constant (no extraction at all)                         -- the floor with the B reads
//   Q1  + A words by ds_read_b64 (one per two fragments) and one v_bfe_u32 per fragment                 -- k_gemm today
//   Q2  + A words by ds_read_b64 and one v_and_b32 (VOP2, field left in place) per fragment
//   Q3  + A operands by ds_read_u8 straight into the low word of the operand pair (byte-expanded genotypes in LDS): no VALU
//   Q4  Q3 + the expansion work of a slab spread over the K-steps: per K-step 2 v_and_b32 + 1/2 ds_write_b128 + 1/8 ds_read_b128
// CC = B fragments per K-step (8: n = 32; 1, 2, 3: the narrow tiles with AA = 16)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
#define SB() __builtin_amdgcn_sched_barrier(0)
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

#define DSU8(dst, addr, off) asm volatile("ds_read_u8 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off) : "memory")
#define DSB64(dst, addr, off) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off) : "memory")
#define DSB128(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off) : "memory")
#define DSW128(addr, src, off) asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(addr), "v"(src), "n"(off) : "memory")
#define WAIT(n) asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(n) : "memory")

template <int Q, int AA, int CC>
__global__ void __launch_bounds__(256, 2) k(double *out, int iters) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 16384; i += 256) reinterpret_cast<uint32_t *>(smem)[i] = 0x01020001u * ((i & 1) + 1) & 0x03030303u;
  for (int i = threadIdx.x; i < 4096; i += 256) reinterpret_cast<double *>(smem + 65536)[i] = 1.0 + 1e-3 * (i & 63);
  __syncthreads();
  double acc[AA][CC];
  for (int g = 0; g < AA; g++) for (int h = 0; h < CC; h++) acc[g][h] = 0;
  u32x2 ap[2][AA];
  for (int s = 0; s < 2; s++) for (int g = 0; g < AA; g++) { uint32_t z; asm volatile("v_mov_b32 %0, 0" : "=v"(z)); ap[s][g].x = 1; ap[s][g].y = z; }
  u32x2 aw[2][AA / 2];
  double bf[2][CC];
  const uint32_t lds0 = (uint32_t)(size_t)(__attribute__((address_space(3))) void *)smem;
  const uint32_t a_addr = lds0 + wave * 4096 + (lane & 3) * 40 + (lane >> 2);             // byte-expanded image: row pitch 40, k along bytes
  const uint32_t w_addr = lds0 + wave * 4096 + (lane & 3) * 32;                            // packed image: 16-lane broadcast
  const uint32_t b_addr = lds0 + 65536 + lane * 8;
  const uint32_t x_addr = lds0 + 32768 + threadIdx.x * 16;
  const uint32_t sh = 2 * (lane >> 2), mask = 3u << sh;
  u32x4 raw = {0x12345678u, 0x9abcdef0u, 0x0f1e2d3cu, 0x4b5a6978u};

  auto issue = [&](int s, int it) {
    if (Q == 5 || Q == 6) {   // two B fragments per ds_read_b128 (fragments interleaved per lane in LDS)
#pragma unroll
      for (int h = 0; h < CC; h += 2) {
        typedef double d2 __attribute__((ext_vector_type(2)));
        d2 t;
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(t) : "v"(b_addr + lane * 8), "n"(0) : "memory");
        bf[s][h] = t.x; bf[s][h + 1] = t.y;
      }
    } else {
#pragma unroll
      for (int h = 0; h < CC; h++) DSB64(bf[s][h], b_addr, h * 512);
    }
    if (Q == 6) {   // A words for four K-steps by one ds_read_b128 per two fragments... modelled as AA/4 b64 reads per step
#pragma unroll
      for (int g = 0; g < AA / 4; g++) DSB64(aw[s][g], w_addr, g * 256);
    }
    if (Q == 1 || Q == 2 || Q == 5) {
#pragma unroll
      for (int g = 0; g < AA / 2; g++) DSB64(aw[s][g], w_addr, g * 256);
    }
    if (Q == 3 || Q == 4) {
#pragma unroll
      for (int g = 0; g < AA; g++) DSU8(ap[s][g].x, a_addr, g * 160);
    }
    if (Q == 4) {   // the expansion of the next slab rides along: per K-step 1/8 raw read, 2 ANDs, 1/2 expanded write
      if ((it & 7) == 0) DSB128(raw, x_addr, 0);
    }
  };
  constexpr int kReads = (Q == 5 || Q == 6 ? CC / 2 : CC) + ((Q == 1 || Q == 2 || Q == 5) ? AA / 2 : 0) + (Q == 6 ? AA / 4 : 0) + ((Q == 3 || Q == 4) ? AA : 0);
  constexpr int kW0 = kReads < 15 ? kReads : 15, kW1 = kReads + 1 < 15 ? kReads + 1 : 15;   // lgkmcnt is a 4-bit counter
  auto compute = [&](int s, int it) {
    if (Q == 1 || Q == 5) {
#pragma unroll
      for (int g = 0; g < AA; g++) ap[s][g].x = __builtin_amdgcn_ubfe((g & 1) ? aw[s][g / 2].y : aw[s][g / 2].x, sh, 2);
    }
    if (Q == 6) {
#pragma unroll
      for (int g = 0; g < AA; g++) ap[s][g].x = __builtin_amdgcn_ubfe((g & 1) ? aw[s][g / 4].y : aw[s][g / 4].x, sh, 2);
    }
    if (Q == 2) {
#pragma unroll
      for (int g = 0; g < AA; g++) asm volatile("v_and_b32 %0, %1, %2" : "=v"(ap[s][g].x) : "v"(mask), "v"((g & 1) ? aw[s][g / 2].y : aw[s][g / 2].x));
    }
    if (Q == 4) {
      u32x4 e;
      asm volatile("v_and_b32 %0, %1, %2" : "=v"(e.x) : "v"(0x03030303u << (2 * (it & 3))), "v"(raw.x));
      asm volatile("v_and_b32 %0, %1, %2" : "=v"(e.y) : "v"(0x03030303u << (2 * (it & 3))), "v"(raw.y));
      e.z = e.x; e.w = e.y;
      if (it & 1) DSW128(x_addr, e, 16384);
    }
    SB();
#pragma unroll
    for (int g = 0; g < AA; g++) {
      const double af = __builtin_bit_cast(double, ap[s][g]);
#pragma unroll
      for (int h = 0; h < CC; h++) acc[g][h] = __builtin_amdgcn_mfma_f64_4x4x4f64(af, bf[s][h], acc[g][h], 0, 0, 0);
    }
    SB();
  };
  issue(0, 0);
  for (int it = 0; it < iters; it += 2) {
    issue(1, it + 1);
    if (Q == 4 && ((it + 1) & 7) == 0) WAIT(kW1); else WAIT(kW0);
    compute(0, it);
    issue(0, it + 2);
    if (Q == 4) WAIT(kW1); else WAIT(kW0);
    compute(1, it + 1);
  }
  WAIT(0);
  double s = 0;
  for (int g = 0; g < AA; g++) for (int h = 0; h < CC; h++) s += acc[g][h];
  for (int g = 0; g < AA; g++) s += ap[0][g].x + ap[1][g].x;
  s += raw.x;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  const int ncu = p.multiProcessorCount; const int iters = 6000;
  double *out; CK(hipMalloc(&out, sizeof(double) * 256 * ncu * 2));
  auto run = [&](const char *name, auto kern, int nmfma) {
    CK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 98304 / 1));
    for (int bpc = 2; bpc <= 2; bpc++) {
      hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
      auto launch = [&] { hipLaunchKernelGGL(kern, dim3(ncu * bpc), dim3(256), 73728, 0, out, iters); };
      launch(); CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0)); for (int i = 0; i < 3; i++) launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 3;
      const double fl = (double)ncu * bpc * 4 * iters * nmfma * 512.0;
      const double cyc = ms * 1e-3 * 2.39e9 / ((double)iters * bpc);
      printf("%-72s blocks/CU=%d %8.3f ms %7.2f TFLOP/s  (%.0f cyc per %d MFMA, ideal %d)\n", name, bpc, ms, fl / ms * 1e-9, cyc, nmfma, nmfma * 16);
    }
  };
#define RUNQ(Q, AA, CC, NAME) run(NAME, k<Q, AA, CC>, AA * CC)
  RUNQ(0, 8, 8, "8x8 Q0: B by ds_read_b64, A constant");
  RUNQ(1, 8, 8, "8x8 Q1: + A words ds_read_b64 + v_bfe_u32 per fragment (today)");
  RUNQ(2, 8, 8, "8x8 Q2: + A words ds_read_b64 + v_and_b32 per fragment");
  RUNQ(3, 8, 8, "8x8 Q3: + A by ds_read_u8 into the operand's low word, no VALU");
  RUNQ(4, 8, 8, "8x8 Q4: Q3 + the slab's expansion work spread over the K-steps");
  RUNQ(5, 8, 8, "8x8 Q5: Q1 with two B fragments per ds_read_b128");
  RUNQ(6, 8, 8, "8x8 Q6: Q5 with half as many A-word reads");
  RUNQ(1, 8, 8, "8x8 Q1 again");
  RUNQ(0, 16, 1, "16x1 Q0");
  RUNQ(1, 16, 1, "16x1 Q1 (today)");
  RUNQ(2, 16, 1, "16x1 Q2 v_and");
  RUNQ(3, 16, 1, "16x1 Q3 ds_read_u8");
  RUNQ(4, 16, 1, "16x1 Q4 ds_read_u8 + expansion");
  RUNQ(0, 16, 2, "16x2 Q0");
  RUNQ(1, 16, 2, "16x2 Q1 (today)");
  RUNQ(3, 16, 2, "16x2 Q3 ds_read_u8");
  RUNQ(4, 16, 2, "16x2 Q4 ds_read_u8 + expansion");
  RUNQ(1, 16, 3, "16x3 Q1 (today)");
  RUNQ(4, 16, 3, "16x3 Q4 ds_read_u8 + expansion");
  RUNQ(1, 16, 4, "16x4 Q1 (today)");
  RUNQ(4, 16, 4, "16x4 Q4 ds_read_u8 + expansion");
  return 0;
}
