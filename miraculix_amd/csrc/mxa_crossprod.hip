// mxa_crossprod.hip -- integer crossproduct M = X * X^T on the matrix cores, exact.
//
// Replaces src/cuda/snp_multiply_cuda.cu:38-382 of the reference (CUTLASS u4 TensorOp GEMM with the two-MMA 2-bit trick
// snp_multiply_cuda.h:121-199, per-tile PCIe re-uploads, host int32->double mirror loop) with a device-resident design:
// X is staged once (2 bits per value, the 256-row x 32-byte tiled layout of mxa_internal.h), every upper-triangular 256x256 tile
// is one workgroup, the packed rows go HBM -> LDS by lane-linear LDS-DMA, each wave unpacks its 2-bit words in registers and the
// epilogue converts to fp64 and writes the tile and its mirror image.  Two exact engines:
//   k_crossprod_f4  (default)  v_mfma_scale_f32_32x32x64_f8f6f4 with FP4 (e2m1) operands: a 2-bit value z in {0..3} in the low bits
//                   of a nibble IS the e2m1 number z/2, so the unpack is 3 VALU per 16 values (two masks and a shift) and the
//                   instruction runs at twice the int8 rate (tools/mfma_f4_probe.hip: exact, 32.8 cycles, 8.3 Pop/s bare loop).
//                   Products are multiples of 1/4; the fp32 accumulator is exact while sum z z' < 2^24, i.e. for K < 1 864 135 with
//                   values up to 3 and K < 4 194 304 when the staged matrix holds no 3 (checked while staging).
//   k_crossprod_i8  (longer K)  v_mfma_i32_32x32x32_i8, 7 VALU per 16 values, exact int32 for K < 2.3e8; the same K-step pipeline.
#include "../../include/miraculix_amd.h"
#include "mxa_internal.h"
#include "mxa_queue.h"
#include "mxa_hostmem.h"
#include <atomic>
#include <chrono>
#include <thread>
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <functional>
#include <mutex>
#include <vector>

namespace mxa {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

using gptr_t = const __attribute__((address_space(1))) void *;
using lptr_t = __attribute__((address_space(3))) void *;

// ---- the reference's PLINK -> 2-bit byte table (snp_multiply_cuda.h:202-210): 00->0, 10->1, 11->2, and a byte that
// holds a missing pair (01) anywhere becomes 0xFF.  SWAR on 4 bytes at a time.
__device__ __forceinline__ uint32_t plink_lut4(uint32_t w) {
  const uint32_t H = (w >> 1) & 0x55555555u, L = w & 0x55555555u;
  const uint32_t z = ((H & L) << 1) | (H & ~L);
  uint32_t miss = L & ~H;                  // bit 2q of a byte set <=> field q is 01
  miss |= miss >> 4;                       // fold fields (0,2) and (1,3); garbage from the next byte lands in bits 4..7
  miss &= 0x05050505u;
  miss |= miss >> 2;
  miss &= 0x01010101u;                     // bit 0 of each byte: the byte holds a missing pair
  return z | (miss * 0xFFu);
}

__global__ void __launch_bounds__(256) k_plink_lut(uint32_t *__restrict__ d, size_t ndwords) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < ndwords; i += (size_t)gridDim.x * blockDim.x) d[i] = plink_lut4(d[i]);
}

// copy rows (src pitch arbitrary) into the tiled layout (mxa_internal.h: byte b of row r -> ((r/256)*nslabs + b/32)*8192 + (r%256)*32
// + b%32), optionally applying the table; padding bytes/rows stay zero (the buffer is memset first).  *has3 |= 1 when a staged field
// holds the value 3 (raw 2-bit input, or a byte with a missing pair under the reference's table).
__global__ void __launch_bounds__(256) k_xstage(const uint8_t *__restrict__ src, size_t src_pitch, long row_bytes, long nrows,
                                                uint8_t *__restrict__ dst, long nslabs, long dst_row0, int apply_lut, int *__restrict__ has3) {
  const long dpr = (row_bytes + 3) / 4;
  const long total = nrows * dpr;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const long r = idx / dpr, d = idx - r * dpr, b = d * 4;
    const uint8_t *p = src + (size_t)r * src_pitch + b;
    uint32_t w = 0, keep = 0;
    if (b + 4 <= row_bytes && (reinterpret_cast<size_t>(p) & 3) == 0) { w = __builtin_nontemporal_load(reinterpret_cast<const uint32_t *>(p)); keep = 0xFFFFFFFFu; }   // one aligned dword (read once)
    else
      for (int u = 0; u < 4; u++)
        if (b + u < row_bytes) { w |= (uint32_t)p[u] << (8 * u); keep |= 0xFFu << (8 * u); }
    if (apply_lut) w = plink_lut4(w) & keep;
    if (w & (w >> 1) & 0x55555555u) atomicOr(has3, 1);
    const long R = dst_row0 + r;
    *reinterpret_cast<uint32_t *>(dst + ((size_t)(R / kTileRows) * nslabs + (size_t)(b / kSlabBytes)) * kTileBytes + (size_t)(R % kTileRows) * kSlabBytes + b % kSlabBytes) = w;
  }
}

// ---- main kernel -----------------------------------------------------------------------------------------------
constexpr int kXT = 256;              // tile edge (rows of X per operand block)
constexpr int kXStageK = 128;         // genotypes per LDS stage = 32 packed bytes per row
constexpr int kXStageBytes = kXStageK / 4;
constexpr int kXOpBytes = kXT * kXStageBytes;     // 8 KiB per operand per stage
constexpr int kXBufBytes = 2 * kXOpBytes;

// 16 two-bit fields of a dword -> 16 int8 in 4 dwords (field order permuted identically for both operands)
__device__ __forceinline__ v4i unpack16(uint32_t w) {
  v4i r;
  r[0] = (int)(w & 0x03030303u);
  r[1] = (int)((w >> 2) & 0x03030303u);
  r[2] = (int)((w >> 4) & 0x03030303u);
  r[3] = (int)((w >> 6) & 0x03030303u);
  return r;
}

// Element-wise map applied by the epilogue (round 3: the GRM / LD post-processing of the reference's binding, crossproduct.jl:83-152, FUSED into
// the crossproduct -- SURVEY.md 8f-3 -- instead of three more passes over the 8 n^2-byte result).  Everything the map needs is known BEFORE the
// product: the column sums of M = X X^T are X (X^T 1) and its diagonal is the row-wise sum of squares, both exact integers computed from the staged
// 2-bit matrix (k_x_colsum, k_x_rowstats).  The same two functions serve the unfused kernels (k_grm_update, k_ld_center / k_ld_scale: kept for
// MXA_XPROD_FUSED_POST=0 and as the bit-identity check of the tests): i = row index, j = column index of the element as stored.
struct XPost {
  const double *u = nullptr;      // GRM: column sums cs of M;  LD: allele frequencies f
  const double *w = nullptr;      // LD: 1 / sigma
  const double *scal = nullptr;   // GRM: scal[0] = sum(cs), scal[1] = 2 sum f (1 - f)
  double a = 0.0;                 // GRM: 1 / n;  LD: 4 * indiv
  int do_scale = 0;
};
// The two divisions of the reference (by the scalar c, by sigma_i and sigma_j) are multiplications by reciprocals formed once (<= 1 ulp from the
// quotient; the stated tolerance of this path is 1e-12): an fp64 division is ~15 instructions on the pipe the epilogue shares with nothing else.
__device__ __forceinline__ double grm_map(double v, double cs_i, double cs_j, double inv_n, double tot_nn, double inv_c, int do_scale) {
  v = fma(-cs_i, inv_n, v);       // BLAS.ger!(-1/indiv, col_sum, one_vector, M)
  v = fma(-cs_j, inv_n, v);       // BLAS.ger!(-1/indiv, one_vector, col_sum, M)
  v = v + tot_nn;                 // M .+= sum(col_sum) / indiv^2
  if (do_scale) v *= inv_c;       // M ./= 2 sum f (1 - f)
  return v;
}
__device__ __forceinline__ double ld_center_map(double v, double f_i, double f_j, double four_indiv) { return fma(-four_indiv * f_i, f_j, v); }   // syr!('U', -4 indiv, f, M)
__device__ __forceinline__ double ld_scale_map(double v, double is_i, double is_j) { return v * is_i * is_j; }                                    // M ./= sigma; M ./= sigma' (is = 1 / sigma)

// Epilogue shared by both engines.  32x32 C/D map: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5); element (gi, gj) = M[gi][gj].
// The output holds columns [c0, ..) of M with leading dimension ld (whole matrix: c0 = 0, ld = n).
// Direct image: M[gj, gi] at ans[gj + (gi-c0)*ld], lanes run along gj (256-byte segments).  Mirror image M[gi, gj] at
// ans[gi + (gj-c0)*ld]: the tile is transposed through a per-wave LDS scratch (row stride 33 doubles: conflict-free both ways)
// so its lanes run along gi as well.  AccT = v16i: exact int32 sums; v16f: sums of z z' / 4 (FP4 engine), exact, times 4.
typedef float v16f __attribute__((ext_vector_type(16)));
// the result is written once and never read by this kernel: non-temporal stores keep the 8 n^2 bytes from displacing the packed operand tiles, which ~n/256
// tiles re-read, in the L2s and the Infinity Cache (MXA_XPROD_NT_STORE=0 at compile time: plain stores, for an A/B; round 3: docs/HISTORY.md)
#ifndef MXA_XPROD_NT_STORE
#define MXA_XPROD_NT_STORE 1
#endif
__device__ __forceinline__ void xstore(double *p, double v) {
#if MXA_XPROD_NT_STORE
  __builtin_nontemporal_store(v, p);
#else
  *p = v;
#endif
}
// POST: 0 plain crossproduct, 1 GRM map, 2 LD map (XPost above); each stored element is mapped with ITS OWN (row, column), so both images equal what
// the unfused element-wise kernels produce.  With a map the 32 x 32 block goes to the LDS scratch first (static accumulator indices) and both images
// are written by loops over IT, never over the accumulators: when the maps still held fp64 divisions, their 512-fold unrolled code exceeded the
// compiler's full-unroll budget, the loops stayed rolled, and a rolled loop over the accumulators indexes them dynamically, i.e. moves them to
// scratch memory for the whole kernel (measured: 10x).  With the reciprocal maps everything unrolls; this form stays safe if it ever does not.
template <typename AccT, int POST>
__device__ __forceinline__ void xprod_store(const AccT (&acc)[4][4], char *smem, int wave, int lane, int wi, int wj, long i0, long j0, int images, long n,
                                            double *__restrict__ ans, long ld, long c0, const XPost &post) {
  double *scratch = reinterpret_cast<double *>(smem) + wave * (32 * 33);   // the DMA ring is dead after the last barrier
  const int col = lane & 31, hh = lane >> 5, rq = 4 * hh;
  constexpr double scale = __is_same(AccT, v16f) ? 4.0 : 1.0;
  if constexpr (POST == 0) {
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
      for (int b = 0; b < 4; b++) {
        const long gi_base = i0 + wi * 128 + a * 32, gj_base = j0 + wj * 128 + b * 32;
        const long gj = gj_base + col;
#pragma unroll
        for (int r = 0; r < 16; r++) {
          const int row = (r & 3) + 8 * (r >> 2) + rq;
          const double v = (double)acc[a][b][r] * scale;
          if ((images & 1) && gi_base + row < n && gj < n) xstore(&ans[(size_t)gj + (size_t)(gi_base + row - c0) * ld], v);
          scratch[row * 33 + col] = v;
        }
        if (images & 2) {
#pragma unroll
          for (int it = 0; it < 16; it++) {
            const int cc = 2 * it + hh;                       // column of the tile = gj offset; lanes (lane&31) run along gi
            const double v = scratch[col * 33 + cc];
            const long gi = gi_base + col, gjj = gj_base + cc;
            if (gi < n && gjj < n) xstore(&ans[(size_t)gi + (size_t)(gjj - c0) * ld], v);
          }
        }
      }
  } else {
    double tot_nn = 0.0, cc_scale = 1.0;
    if (POST == 1) { tot_nn = post.scal[0] / ((double)n * (double)n); cc_scale = post.do_scale ? 1.0 / post.scal[1] : 1.0; }
    auto map = [&](double v, long i, long j) -> double {       // element M[i][j] (i, j < n)
      if (POST == 1) return grm_map(v, post.u[i], post.u[j], post.a, tot_nn, cc_scale, post.do_scale);
      return ld_scale_map(ld_center_map(v, post.u[i], post.u[j], post.a), post.w[i], post.w[j]);
    };
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
      for (int b = 0; b < 4; b++) {
        const long gi_base = i0 + wi * 128 + a * 32, gj_base = j0 + wj * 128 + b * 32;
        const long gj = gj_base + col, gi = gi_base + col;
#pragma unroll
        for (int r = 0; r < 16; r++) scratch[((r & 3) + 8 * (r >> 2) + rq) * 33 + col] = (double)acc[a][b][r] * scale;
        if ((images & 1) && gj < n) {                          // direct image: row index gj, column index gi_base + row
#pragma unroll
          for (int r = 0; r < 16; r++) {
            const int row = (r & 3) + 8 * (r >> 2) + rq;
            if (gi_base + row < n) xstore(&ans[(size_t)gj + (size_t)(gi_base + row - c0) * ld], map(scratch[row * 33 + col], gj, gi_base + row));
          }
        }
        if ((images & 2) && gi < n) {                          // mirror image: row index gi, column index gj_base + cc
#pragma unroll
          for (int it = 0; it < 16; it++) {
            const int cc = 2 * it + hh;
            if (gj_base + cc < n) xstore(&ans[(size_t)gi + (size_t)(gj_base + cc - c0) * ld], map(scratch[col * 33 + cc], gi, gj_base + cc));
          }
        }
      }
  }
}

// Both engines: 4 waves, one per SIMD, wave tile 128 x 128 (16 accumulator tiles = 256 registers), one workgroup per CU.  (History: 8 waves with
// 128 x 64 wave tiles stalled at 43-54 % of the int8 peak on their unpack VALU density; the round-1/2 int8 kernel k_crossprod2 -- 3-deep ring, the
// packed words of a whole stage prefetched mid-stage -- ran 2704 cycles per stage of 2048 ideal at 2.38 GHz, issue-bound; round 3 moved the int8
// engine onto the FP4 kernel's K-step pipeline below: 2272 cycles at 2.19 GHz, now power-bound like the FP4 engine.)
__device__ __forceinline__ void xdma16_s(const void *sbase, uint32_t voff, uint32_t lds_addr) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %0" ::"s"(sbase), "v"(voff), "s"(lds_addr) : "memory", "m0");
}

// ---- FP4 engine ----------------------------------------------------------------------------------------------------------------
// Per stage of 128 genotypes TWO K-steps of 64 (v_mfma_scale_f32_32x32x64_f8f6f4, FP4
// operands, unit scales).  Lane (row r = lane&31, K half h = lane>>5) reads 16 bytes = 64 genotypes of its row per stage; K-step ks uses
// dwords 2ks, 2ks+1 of them.  A dword of 16 two-bit values z becomes two dwords of 8 nibbles 00zz -- the e2m1 numbers z/2 -- by
// (w & 0x33333333) and ((w >> 2) & 0x33333333): 6 VALU per fragment of 32 values, 48 per K-step of 16 MFMAs (the int8 engine: 56 VALU per
// 16 MFMAs of HALF the K).  The K order inside a fragment is permuted identically for both operands, so the dot product is unchanged.
typedef int v8i __attribute__((ext_vector_type(8)));
constexpr int kF4Bufs = 8;                        // ring depth of the FP4 kernel
constexpr int kF4Lds = kF4Bufs * kXBufBytes;      // 128 KiB (one workgroup per CU: 256 accumulator registers per lane)
__device__ __forceinline__ v4i unpack_f4(uint32_t w0, uint32_t w1) {
  v4i r;
  r[0] = (int)(w0 & 0x33333333u);
  r[1] = (int)((w0 >> 2) & 0x33333333u);
  r[2] = (int)(w1 & 0x33333333u);
  r[3] = (int)((w1 >> 2) & 0x33333333u);
  return r;
}
__device__ __forceinline__ v16f mfma_f4(const v4i &a, const v4i &b, const v16f &c) {
  const v8i a8 = {a[0], a[1], a[2], a[3], 0, 0, 0, 0}, b8 = {b[0], b[1], b[2], b[3], 0, 0, 0, 0};   // FP4 operands occupy 4 of the 8 registers
  return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, c, 4, 4, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);   // cbsz = blgp = 4: e2m1; scales 2^0
}

// EXP (diagnostic instantiations only, results are wrong): 1 = no unpack VALU; 2 = no DMA / barrier / LDS reads inside the loop;
// 3 = barrier only; 4 = DMA + LDS reads without the barrier; 5 = barrier + DMA, no LDS reads; 6 = barrier + LDS reads, no DMA
// The same pipeline serves the int8 engine (I8, round 3): a "K-step of 64" is then two v_mfma_i32_32x32x32_i8 sub-steps -- the lane's two dwords of
// a K-step unpack into two int8 fragments (14 VALU per sub-block instead of 6), 8 MFMAs per sub-block row instead of 4, exact int32 sums.  It replaced
// k_crossprod2 (3-deep ring, words of a whole stage prefetched mid-stage: 2704 cycles per stage of 2048 ideal at 2.38 GHz -- issue-bound, not
// power-bound).
struct FragI8 { v4i lo, hi; };
template <bool I8> struct XFrag { using type = v4i; using acc = v16f; };
template <> struct XFrag<true> { using type = FragI8; using acc = v16i; };

// second meeting point of a gang (round 4; MXA_XPROD_GANG_MID = parts of the K range: 1 = no meeting, 2 = one meeting half way, the default): the tiles of a
// gang start together but drift apart inside a 2.6 ms tile; half way through the K range every member adds to the gang's counter and waits for the others --
// bounded by the join time and by 4 % of the tile's own duration, whichever is shorter -- so that the second half streams in step again
struct GangMid { int *ctr; int target; unsigned ticks; int parts; int stride; };   // parts - 1 meetings inside a tile, meeting q uses ctr[(q - 1) * stride]
template <bool DIAG, int EXP, bool I8, int POST>
__device__ __forceinline__ void xprod_tile(const uint8_t *__restrict__ X, long nslabs, int stages, const int4 t, size_t tile_index, long n, double *__restrict__ ans,
                                           long ld, long c0, unsigned long long *__restrict__ diag, const XPost &post, const GangMid gm = GangMid{nullptr, 0, 0, 1, 0}) {
  using FragT = typename XFrag<I8>::type;
  using AccT = typename XFrag<I8>::acc;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wi = wave >> 1, wj = wave & 1;
  const long i0 = (long)t.x * kXT, j0 = (long)t.y * kXT;
  const uint32_t lds0 = (uint32_t)(size_t)(lptr_t)smem;
  const uint32_t v_lane = (uint32_t)lane * 16;
  const char *XI = reinterpret_cast<const char *>(X) + (size_t)t.x * nslabs * kTileBytes;
  const char *XJ = reinterpret_cast<const char *>(X) + (size_t)t.y * nslabs * kTileBytes;
  auto issue = [&](int stage, int buf) {
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const int u = wave + 4 * i;
      const int op = u >> 3, uu = u & 7;
      xdma16_s((op ? XJ : XI) + (size_t)stage * kTileBytes + uu * 1024, v_lane, lds0 + buf * kXBufBytes + op * kXOpBytes + uu * 1024);
    }
  };

  AccT acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; a++)
#pragma unroll
    for (int b = 0; b < 4; b++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[a][b][r] = 0;

  const int a_off = (wi * 128 + (lane & 31)) * kXStageBytes + (lane >> 5) * 16;
  const int b_off = kXOpBytes + (wj * 128 + (lane & 31)) * kXStageBytes + (lane >> 5) * 16;

  // Ring of NB buffers (a stage lasts only ~0.5 us at the FP4 rate).  Stages 0 .. NB-1 are in flight at the start; at the start of stage s
  // the words of stage s are already in registers, so its buffer takes stage s + NB, and stage s + 1 is waited for (stages s+2 .. s+NB-1 stay
  // in flight).
  constexpr int NB = kF4Bufs;
#pragma unroll
  for (int i = 0; i < NB; i++) issue(min(i, stages - 1), i);   // always NB stages in flight (clamped: short K re-loads the last stage)
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * (NB - 1)) : "memory");
  __syncthreads();
  unsigned long long t0 = 0, r0 = 0;
  if (DIAG) { t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }
  // Software pipeline in K-steps t = 2 s + ks (a stage has two K-steps of 64 genotypes):
  //   during K-step t the wave (1) issues the ds_read_b64 of the packed words of K-step t+2 into word set W[t%2], (2) unpacks the words of
  //   K-step t+1 (read during t-1, word set W[(t+1)%2]) into fragment set F[(t+1)%2], 3 VALU after every MFMA, and (3) issues the 16 MFMAs
  //   of K-step t from fragment set F[t%2].  Neither an LDS latency nor a VALU result is ever waited for, and no register set is copied
  //   (the period of the alternation, two K-steps, is exactly one stage, so the stage loop needs no unrolling).
  //   Stage start: the words of stage s were all read during stage s-1, so after the barrier buffer s%NB takes stage s+NB; stage s+1
  //   must have landed (its words are read during this stage); stages s+2 .. s+NB-1 stay in flight.
  typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
  u32x2 wa0[4], wb0[4], wa1[4], wb1[4];      // W[0] / W[1]: words of an even / odd K-step, 4 A and 4 B sub-blocks of 32 rows
  FragT fa0[4], fb0[4], fa1[4], fb1[4];      // F[0] / F[1]
  auto unpack = [](const u32x2 &w) -> FragT {
    if constexpr (I8) { FragI8 r; r.lo = unpack16(w.x); r.hi = unpack16(w.y); return r; }
    else {
      if (EXP == 1) { v4i r = {(int)w.x, (int)w.y, (int)w.x, (int)w.y}; return r; }
      return unpack_f4(w.x, w.y);
    }
  };
  auto mma = [](const FragT &fa, const FragT &fb, const AccT &c) -> AccT {
    if constexpr (I8) return __builtin_amdgcn_mfma_i32_32x32x32_i8(fa.hi, fb.hi, __builtin_amdgcn_mfma_i32_32x32x32_i8(fa.lo, fb.lo, c, 0, 0, 0), 0, 0, 0);
    else return mfma_f4(fa, fb, c);
  };
#pragma unroll
  for (int a = 0; a < 4; a++) {
    wa0[a] = *reinterpret_cast<const u32x2 *>(smem + a_off + a * 32 * kXStageBytes);
    wb0[a] = *reinterpret_cast<const u32x2 *>(smem + b_off + a * 32 * kXStageBytes);
    wa1[a] = *reinterpret_cast<const u32x2 *>(smem + a_off + a * 32 * kXStageBytes + 8);
    wb1[a] = *reinterpret_cast<const u32x2 *>(smem + b_off + a * 32 * kXStageBytes + 8);
  }
#pragma unroll
  for (int a = 0; a < 4; a++) { fa0[a] = unpack(wa0[a]); fb0[a] = unpack(wb0[a]); fa1[a] = fa0[a]; fb1[a] = fb0[a]; }
  int buf = 0;
  // K-step: MFMAs from (FA, FB); unpack (WAU, WBU) -> (FAN, FBN); read the words at LDS address RD (+ sub-block) into (WAR, WBR).
  // Everything that is not an MFMA is spread over the four groups of 4 MFMAs (one DMA unit when DO_DMA, two ds_read_b64, 12 unpack VALU per
  // group) and the whole stage is ONE basic block (no branches: the DMA of the last NB stages re-loads the final stage into buffers nobody
  // reads, the reads after the last stage hit a valid buffer), so that the instruction scheduler keeps the interleave it is given:
  // clustered at the stage start the 4 DMA issues and 16 LDS reads cost ~165 + ~140 cycles of a 1024-cycle stage (profiles/r02_mfma_f4_probe.txt).
#define MXA_F4_KSTEP(FA, FB, WAU, WBU, FAN, FBN, WAR, WBR, RD, DO_DMA)                                                                      \
  {                                                                                                                                        \
    _Pragma("unroll") for (int a = 0; a < 4; a++) {                                                                                        \
      if (DO_DMA && EXP != 2 && EXP != 3 && EXP != 6) {                                                                                    \
        const int u = wave + 4 * a, op = u >> 3, uu = u & 7;                                                                               \
        xdma16_s((op ? XJ : XI) + (size_t)dma_stage * kTileBytes + uu * 1024, v_lane, lds0 + dma_buf * kXBufBytes + op * kXOpBytes + uu * 1024); \
      }                                                                                                                                    \
      if (EXP != 2 && EXP != 3 && EXP != 5) {                                                                                              \
        WAR[a] = *reinterpret_cast<const u32x2 *>((RD) + a_off + a * 32 * kXStageBytes);                                                   \
        WBR[a] = *reinterpret_cast<const u32x2 *>((RD) + b_off + a * 32 * kXStageBytes);                                                   \
      }                                                                                                                                    \
      FAN[a] = unpack(WAU[a]);                                                                                                             \
      FBN[a] = unpack(WBU[a]);                                                                                                             \
      _Pragma("unroll") for (int b = 0; b < 4; b++) acc[a][b] = mma(FA[a], FB[b], acc[a][b]);                                              \
      if constexpr (I8) {   /* 8 MFMAs, 28 unpack VALU, one DMA unit, two LDS reads */                                                     \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x020, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, 4, 0); \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, 4, 0); \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, 4, 0); \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);                              \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);                              \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);                              \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);                              \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);                              \
      } else {                                                                                                                             \
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x020, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, 3, 0); \
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, 3, 0); \
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, 3, 0); \
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);                                \
      }                                                                                                                                    \
      __builtin_amdgcn_sched_barrier(0);                                                                                                   \
    }                                                                                                                                      \
  }
  // two halves of the K range with the gang's second meeting between them (gm.ctr == nullptr: one pass); the stage loop itself stays one basic block
  const int nparts = gm.ctr ? gm.parts : 1;
  for (int part = 0; part < nparts; part++) {
  if (part > 0) {
    if (threadIdx.x == 0) {
      int *c = gm.ctr + (size_t)(part - 1) * gm.stride;
      __hip_atomic_fetch_add(c, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned long long tm = __builtin_amdgcn_s_memrealtime();
      while (__hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < gm.target && __builtin_amdgcn_s_memrealtime() - tm < gm.ticks) __builtin_amdgcn_s_sleep(2);
    }
    __syncthreads();
  }
  const int s_lo = (int)((long)stages * part / nparts), s_hi = (int)((long)stages * (part + 1) / nparts);
  for (int s = s_lo; s < s_hi; s++) {
    // stage start: stage s+1 must have landed -- exactly (NB-2) stages' DMAs may stay in flight (one stage's 4 units are issued per stage,
    // always); this wave's reads of buffer s%NB were issued a K-step ago and are complete; after the barrier that buffer is refilled
    if (EXP != 2) {
      if (EXP != 3 && EXP != 6) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * (NB - 2)) : "memory");
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (EXP != 4) __syncthreads();
    }
    const int dma_stage = min(s + NB, stages - 1), dma_buf = buf;
    buf = buf == NB - 1 ? 0 : buf + 1;
    const char *nxt = smem + buf * kXBufBytes;
    // t = 2s:   MFMAs F[0]; unpack W[1] (K-step 1 of this stage) -> F[1]; read K-step 0 of stage s+1 -> W[0]; refill buffer s%NB
    MXA_F4_KSTEP(fa0, fb0, wa1, wb1, fa1, fb1, wa0, wb0, nxt, true)
    // t = 2s+1: MFMAs F[1]; unpack W[0] (K-step 0 of stage s+1) -> F[0]; read K-step 1 of stage s+1 -> W[1]
    MXA_F4_KSTEP(fa1, fb1, wa0, wb0, fa0, fb0, wa1, wb1, nxt + 8, false)
  }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the redundant refills of the last stages have landed before the ring becomes scratch
#undef MXA_F4_KSTEP
  __syncthreads();   // all waves are done with the ring before it is reused as the epilogue scratch
  if (DIAG) {
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0 && diag) { diag[2 * tile_index] = t1 - t0; diag[2 * tile_index + 1] = r1 - r0; }
  }
  xprod_store<AccT, POST>(acc, smem, wave, lane, wi, wj, i0, j0, t.z, n, ans, ld, c0, post);
}

template <bool DIAG, int EXP = 0, int POST = 0>
__global__ void __launch_bounds__(256, 1)
k_crossprod_f4(const uint8_t *__restrict__ X, long nslabs, int stages, const int4 *__restrict__ tiles, long n, double *__restrict__ ans,
               long ld, long c0, unsigned long long *__restrict__ diag, XPost post) {
  const int4 t = tiles[blockIdx.x];
  if (t.z == 0) return;                         // padding entry of the XCD-aware tile order (whole workgroup, before any barrier)
  xprod_tile<DIAG, EXP, false, POST>(X, nslabs, stages, t, blockIdx.x, n, ans, ld, c0, diag, post);
}
template <bool DIAG, int POST = 0>
__global__ void __launch_bounds__(256, 1)
k_crossprod_i8(const uint8_t *__restrict__ X, long nslabs, int stages, const int4 *__restrict__ tiles, long n, double *__restrict__ ans,
               long ld, long c0, unsigned long long *__restrict__ diag, XPost post) {
  const int4 t = tiles[blockIdx.x];
  if (t.z == 0) return;
  xprod_tile<DIAG, 0, true, POST>(X, nslabs, stages, t, blockIdx.x, n, ans, ld, c0, diag, post);
}

// ---- gang-synchronised persistent form (round 3) ------------------------------------------------------------------------------------------
// Every 256-row block of X is an operand of ~n/256 tiles, and the kernel above lets the tiles of an XCD drift apart along K (a finished workgroup is
// replaced at once, at its own time), so concurrent tiles rarely find each other's rows in the XCD's L2: the counters show 3.0 TB fetched per config-3
// launch (12.5 GB of operand, 243x), 3.7 TB/s next to a power-bound MFMA stream -- with every tile reading the SAME two blocks (results wrong) the
// clock rises from 2.01 to 2.19 GHz and the launch takes 7 % less.  Here ONE workgroup per CU stays resident and the P workgroups of an XCD advance
// through that XCD's tile list in GANGS: a workgroup that has finished claims the next slot (one returning atomic add) and waits until all P slots of
// the gang are claimed, i.e. until the whole XCD is ready, so that the gang's tiles -- 4 x 8 tiles of a super-tile: 12 row blocks for 32 tiles --
// start together and stream their operands in step (launches cut into waves of 256 tiles, the same synchronisation by other means, fetch 1.13 TB
// instead of 3.0).  Every wait is bounded by the clock (s_memrealtime): a workgroup that is not joined in time goes on alone, so the grid always
// drains whatever the hardware did with the workgroups (fewer CUs, another process on the chip); only the sharing is lost then.
// gang[0..7]: claim counters of the XCD lists, gang[8..15]: workgroups seen per XCD, gang[16]: workgroups seen in total (zeroed by the launcher).
constexpr int kGangCtrs = 17;
constexpr unsigned long long kGangJoinTicks = 10000;      // 100 us at the 100 MHz of s_memrealtime: ~4 % of a config-3 tile
constexpr unsigned long long kGangStartTicks = 200000;    // 2 ms for the whole grid to become resident
template <bool I8, int POST>
__global__ void __launch_bounds__(256, 1)
k_crossprod_gang(const uint8_t *__restrict__ X, long nslabs, int stages, const int4 *__restrict__ tiles, int slots_per_xcd, long n, double *__restrict__ ans,
                 long ld, long c0, XPost post, int *__restrict__ gang, unsigned join_ticks, int xcc_mask, int *__restrict__ mid, int mid_parts) {
  __shared__ int sh_val;
  const int xcc = hw_xcc_id() & xcc_mask;             // mask 7; the tests narrow it to emulate a chip that populates fewer XCDs (the other lists are then stolen)
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(gang + 8 + xcc, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_fetch_add(gang + 16, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__hip_atomic_load(gang + 16, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < (int)gridDim.x && __builtin_amdgcn_s_memrealtime() - t0 < kGangStartTicks)
      __builtin_amdgcn_s_sleep(16);
    sh_val = __hip_atomic_load(gang + 8 + xcc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  const int P = max(1, sh_val);                       // workgroups of this XCD = gang size
  __syncthreads();
  __shared__ int sh_list, sh_own;
  int phase = 0;                                      // lane 0 only: 0 = the own XCD's list (in gangs), 1..7 = the other XCDs' lists once the own one is empty
  for (;;) {                                          // (no waiting there: correctness must not depend on which XCDs the hardware populated, e.g. a partitioned chip)
    if (threadIdx.x == 0) {
      int slot = -1, y = xcc;
      while (phase < 8) {
        y = (xcc + phase) & 7;
        slot = __hip_atomic_fetch_add(gang + y, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (slot < slots_per_xcd) break;
        slot = -1; phase++;
      }
      if (slot >= 0 && phase == 0) {
        const int target = min((slot / P + 1) * P, slots_per_xcd);
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        while (__hip_atomic_load(gang + xcc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target && __builtin_amdgcn_s_memrealtime() - t0 < join_ticks)
          __builtin_amdgcn_s_sleep(4);
      }
      sh_val = slot; sh_list = y; sh_own = (slot >= 0 && phase == 0) ? 1 : 0;
    }
    __syncthreads();
    const int slot = __builtin_amdgcn_readfirstlane(sh_val), list = __builtin_amdgcn_readfirstlane(sh_list);
    if (slot < 0) break;                               // wave-uniform: every list is empty, the whole workgroup leaves
    const int4 tv = tiles[(size_t)8 * slot + list];
    const int4 t = make_int4(__builtin_amdgcn_readfirstlane(tv.x), __builtin_amdgcn_readfirstlane(tv.y), __builtin_amdgcn_readfirstlane(tv.z), 0);   // scalar: the DMA bases live in SGPRs
    // second meeting point (mid != nullptr): the members of a gang of the own list meet again half way through the tile
    GangMid gm{nullptr, 0, 0, 1, 0};
    if (mid && __builtin_amdgcn_readfirstlane(sh_own)) {
      // (everything the stage loop's bounds depend on must be provably wave-uniform: the DMA bases and LDS addresses live in SGPRs)
      const int Pu = __builtin_amdgcn_readfirstlane(P), g = slot / Pu, members = __builtin_amdgcn_readfirstlane(min((g + 1) * Pu, slots_per_xcd) - g * Pu);
      const int idx = __builtin_amdgcn_readfirstlane(xcc * slots_per_xcd + g);
      // the meeting's wait: at most the join bound, and at most ~4 % of this tile (stages x 0.66 us at the FP4 rate = 66 ticks of 10 ns per stage)
      const unsigned mid_ticks = min(join_ticks, (unsigned)(2.64f * (float)stages));
      gm = GangMid{mid + idx, members, mid_ticks, mid_parts, 8 * slots_per_xcd};
      if (!t.z && threadIdx.x == 0)   // a padding entry never reaches the meetings: counted here
        for (int q = 0; q + 1 < mid_parts; q++) __hip_atomic_fetch_add(gm.ctr + (size_t)q * gm.stride, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (t.z) xprod_tile<false, 0, I8, POST>(X, nslabs, stages, t, 0, n, ans, ld, c0, nullptr, post, gm);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the stores of the epilogue: the DMA bookkeeping of the next tile starts from an empty counter
    __syncthreads();                                   // ... and the LDS scratch of the epilogue is free (sh_val / sh_list are rewritten only after this barrier)
  }
}

int launch_plink_lut(uint8_t *d, size_t nbytes, hipStream_t s) {
  const size_t nd = nbytes / 4;
  const int grid = (int)std::min<size_t>((nd + 255) / 256, 256 * 32);
  hipLaunchKernelGGL(k_plink_lut, dim3(grid), dim3(256), 0, s, reinterpret_cast<uint32_t *>(d), nd);
  MXA_HIP(hipGetLastError());
  return 0;
}

// XCD-aware order of a tile list.  Consecutive workgroups are dealt round-robin to the 8 XCDs (own L2 each).  In plain i-major
// order the 32 tiles an XCD works on at a time share one I block and need 32 different J blocks; here the tiles are grouped into
// super-tiles of sr x 8 tiles (sr tile rows, 8 tile columns), whole super-tiles go to the XCD with the shortest list so far, and the
// lists are interleaved (list index = 8 * slot + xcd, padded with no-op entries {0,0,0,0}) so that an XCD streams sr + 8 row blocks
// for 8 * sr tiles.  MXA_XPROD_XCD=0 keeps the plain order (A/B measurement).
// Returns true when the list has the interleaved per-XCD form (false: left in plain order).
static bool xcd_order_tiles(std::vector<int4> &tiles, int nb, int sr) {
  constexpr bool on = true;
  if (!on || tiles.size() < 8 * 64 || sr < 1) return false;
  int i_min = tiles[0].x;
  for (const int4 &t : tiles) i_min = std::min(i_min, t.x);
  const int ncb = (nb + 7) / 8;
  std::vector<std::vector<int4>> super((size_t)((nb - i_min + sr - 1) / sr) * ncb);
  for (const int4 &t : tiles) super[(size_t)((t.x - i_min) / sr) * ncb + t.y / 8].push_back(t);
  std::vector<std::vector<int4>> per_xcd(8);
  for (auto &st : super) {
    if (st.empty()) continue;
    int best = 0;
    for (int x = 1; x < 8; x++) if (per_xcd[x].size() < per_xcd[best].size()) best = x;
    per_xcd[best].insert(per_xcd[best].end(), st.begin(), st.end());
  }
  size_t longest = 0;
  for (auto &v : per_xcd) longest = std::max(longest, v.size());
  std::vector<int4> inter;
  inter.reserve(longest * 8);
  for (size_t slot = 0; slot < longest; slot++)
    for (int x = 0; x < 8; x++) inter.push_back(slot < per_xcd[x].size() ? per_xcd[x][slot] : make_int4(0, 0, 0, 0));
  tiles.swap(inter);
  return true;
}

// Order for the gang-synchronised kernel: the list is cut into GANGS of 32 tiles (the workgroups of an XCD) that are compact in the tile grid -- bands of 4
// tile rows, walked column by column, so a gang is 4 x 8 tiles = 12 row blocks (a little more where it meets the diagonal or the end of a band) -- and the
// gangs are dealt whole to the 8 XCD lists (neighbouring gangs of a band run on different XCDs at the same time and share the band's 4 row blocks through
// the Infinity Cache).  Gang boundaries stay aligned with multiples of 32 slots in every list; with the 8 x 8 super-tiles of xcd_order_tiles one
// partial super-tile (36 tiles on the diagonal) shifts every later gang of that list across two super-tile halves.
static bool gang_order_tiles(std::vector<int4> &tiles) {
  constexpr bool on = true;
  constexpr size_t kGang = 32;
  if (!on || tiles.size() < 8 * 64) return false;
  std::sort(tiles.begin(), tiles.end(), [](const int4 &a, const int4 &b) {
    if (a.x / 4 != b.x / 4) return a.x / 4 < b.x / 4;
    if (a.y != b.y) return a.y < b.y;
    return a.x < b.x;
  });
  std::vector<std::vector<int4>> per_xcd(8);
  size_t next = 0;
  for (size_t g0 = 0; g0 < tiles.size(); g0 += kGang) {
    std::vector<int4> &l = per_xcd[next];
    next = (next + 1) & 7;
    const size_t g1 = std::min(tiles.size(), g0 + kGang);
    l.insert(l.end(), tiles.begin() + (long)g0, tiles.begin() + (long)g1);
    l.resize((l.size() + kGang - 1) / kGang * kGang, make_int4(0, 0, 0, 0));      // only the very last gang is short
  }
  size_t longest = 0;
  for (auto &v : per_xcd) longest = std::max(longest, v.size());
  std::vector<int4> inter;
  inter.reserve(longest * 8);
  for (size_t slot = 0; slot < longest; slot++)
    for (int x = 0; x < 8; x++) inter.push_back(slot < per_xcd[x].size() ? per_xcd[x][slot] : make_int4(0, 0, 0, 0));
  tiles.swap(inter);
  return true;
}

// ---- host side ------------------------------------------------------------------------------------------------------------------
static thread_local bool tl_xprod_shared_device = false;   // set by the panel workers of snp_multiply_gpu when several panels share a device
static std::mutex g_xprof_mutex;   // panels of one call run in several threads (MIRACULIX_NUM_GPUS): the profile counters are shared
namespace {
struct XEvent {   // RAII: events, streams and device buffers are released on every exit path
  hipEvent_t e = nullptr;
  ~XEvent() { if (e) (void)hipEventDestroy(e); }
  int create(unsigned flags = hipEventDefault) { MXA_HIP(hipEventCreateWithFlags(&e, flags)); return 0; }
};
struct XStream {
  hipStream_t s = nullptr;
  ~XStream() { if (s) (void)hipStreamDestroy(s); }
  int create(unsigned flags) { MXA_HIP(hipStreamCreateWithFlags(&s, flags)); return 0; }
};
struct XBuf {
  void *p = nullptr;
  ~XBuf() { if (p) (void)hipFree(p); }
  int alloc(size_t bytes) { MXA_HIP(hipMalloc(&p, bytes ? bytes : 1)); return 0; }
  void release() { if (p) { (void)hipFree(p); p = nullptr; } }
};
}  // namespace

// one launch over a tile list with either engine (f4: FP4 MFMA, else int8 MFMA); diag_out: in-kernel clocks of the DIAG instantiation
// gang_mid_capacity: ints available behind d_gang[32] for the per-gang counters of the second meeting point (0: none)
static int launch_tiles(bool f4, size_t ntiles, hipStream_t s, const uint8_t *d_X, long nslabs, int stages, const int4 *d_tiles, long rows, double *d_ans, long ld,
                        long c0, unsigned long long *d_diag, int post_kind = 0, const XPost &post = XPost(), int *d_gang = nullptr, size_t gang_mid_capacity = 0) {
  static unsigned long long m2 = 0, m3 = 0;   // per-device function attributes
  if (ensure_dyn_lds(reinterpret_cast<const void *>(&k_crossprod_f4<false>), kF4Lds, &m2) || ensure_dyn_lds(reinterpret_cast<const void *>(&k_crossprod_f4<true>), kF4Lds, &m3)) return 1;
  const dim3 grid((unsigned)ntiles), block(256);
  // gang-synchronised persistent form: needs the interleaved per-XCD lists (d_gang != nullptr) and is not instrumented (MXA_DIAG keeps the classic kernels)
  // It pays when the launch is long enough for the power limit to matter and a tile long enough to carry the meeting: config 3 -5 ... -8 %, 30 000 rows x
  // 500k -3.4 %, but K = 50 000 (0.26 ms per tile) +2 ... +8 % at 8 192 - 40 000 rows.  MXA_XPROD_GANG: 0 never, 1 (default) by this estimate, 2 whenever possible.
  static const int gang_on = [] { const char *e = getenv("MXA_XPROD_GANG"); return e ? atoi(e) : 1; }();
  int dev = 0, cus = 0;
  // not when this call is one of several panels computed side by side on ONE device (MIRACULIX_NUM_GPUS above the device count): the gang form wants one
  // workgroup per CU resident at once and would spend its 2 ms start-up wait on workgroups that cannot become resident beside the other panel's kernel
  if (gang_on && d_gang && !d_diag && ntiles % 8 == 0 && !(tl_xprod_shared_device && gang_on < 2)) {
    MXA_HIP(hipGetDevice(&dev));
    MXA_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
  }
  const double est_ms = cus > 0 ? (double)stages * (f4 ? 0.66e-3 : 1.0e-3) * ((double)ntiles / cus) : 0.0;
  if (cus > 0 && (gang_on >= 2 || (stages >= 1024 && est_ms >= 20.0))) {
    const int slots = (int)(ntiles / 8);
    const dim3 pgrid((unsigned)std::max(8, std::min<int>(cus, (int)ntiles)));
    MXA_HIP(hipMemsetAsync(d_gang, 0, sizeof(int) * kGangCtrs, s));
    // second meeting point: one counter per gang and XCD list, behind the 17 control counters when the caller's buffer has room for them
    // MXA_XPROD_GANG_MID = number of parts a tile's K range is cut into (meetings = parts - 1): 0 / 1 none, 2 (default) one meeting half way through.
    // The gang form is only taken for long tiles (stages >= 1024, i.e. K >= 131k: tiles of >= 0.7 ms), and the meeting's wait is bounded by 4 % of the tile.
    const char *e_mid = getenv("MXA_XPROD_GANG_MID");
    const int mid_parts = std::max(1, std::min(8, e_mid ? atoi(e_mid) : 2));
    int *d_mid = nullptr;
    if (mid_parts > 1 && gang_mid_capacity >= (size_t)8 * slots * (mid_parts - 1)) {
      d_mid = d_gang + 32;
      MXA_HIP(hipMemsetAsync(d_mid, 0, sizeof(int) * (size_t)8 * slots * (mid_parts - 1), s));
    }
    static unsigned long long g0 = 0, g1 = 0, g2 = 0, g3 = 0, g4 = 0, g5 = 0;
    static const int xcc_mask = [] { const char *e = getenv("MXA_XPROD_GANG_XCC_MASK"); return e ? atoi(e) & 7 : 7; }();
    static const unsigned join_ticks = [] { const char *e = getenv("MXA_XPROD_GANG_US"); return e ? (unsigned)std::max(0, atoi(e)) * 100u : (unsigned)kGangJoinTicks; }();
#define MXA_GANG_LAUNCH(I8, POST, MASK)                                                                                                          \
    {                                                                                                                                            \
      if (ensure_dyn_lds(reinterpret_cast<const void *>(&k_crossprod_gang<I8, POST>), kF4Lds, &MASK)) return 1;                                   \
      hipLaunchKernelGGL((k_crossprod_gang<I8, POST>), pgrid, block, kF4Lds, s, d_X, nslabs, stages, d_tiles, slots, rows, d_ans, ld, c0, post, d_gang, join_ticks, xcc_mask, d_mid, mid_parts); \
    }
    if (f4) { if (post_kind == 1) MXA_GANG_LAUNCH(false, 1, g1) else if (post_kind == 2) MXA_GANG_LAUNCH(false, 2, g2) else MXA_GANG_LAUNCH(false, 0, g0) }
    else { if (post_kind == 1) MXA_GANG_LAUNCH(true, 1, g4) else if (post_kind == 2) MXA_GANG_LAUNCH(true, 2, g5) else MXA_GANG_LAUNCH(true, 0, g3) }
#undef MXA_GANG_LAUNCH
    MXA_HIP(hipGetLastError());
    return 0;
  }
  if (post_kind == 1 || post_kind == 2) {   // GRM / LD map fused into the epilogue (never with the diagnostic instantiations)
    static unsigned long long p1 = 0, p2 = 0, p3 = 0, p4 = 0;
    if (ensure_dyn_lds(reinterpret_cast<const void *>(&k_crossprod_f4<false, 0, 1>), kF4Lds, &p1) || ensure_dyn_lds(reinterpret_cast<const void *>(&k_crossprod_f4<false, 0, 2>), kF4Lds, &p2) ||
        ensure_dyn_lds(reinterpret_cast<const void *>(&k_crossprod_i8<false, 1>), kF4Lds, &p3) || ensure_dyn_lds(reinterpret_cast<const void *>(&k_crossprod_i8<false, 2>), kF4Lds, &p4)) return 1;
    if (f4 && post_kind == 1) hipLaunchKernelGGL((k_crossprod_f4<false, 0, 1>), grid, block, kF4Lds, s, d_X, nslabs, stages, d_tiles, rows, d_ans, ld, c0, nullptr, post);
    else if (f4) hipLaunchKernelGGL((k_crossprod_f4<false, 0, 2>), grid, block, kF4Lds, s, d_X, nslabs, stages, d_tiles, rows, d_ans, ld, c0, nullptr, post);
    else if (post_kind == 1) hipLaunchKernelGGL((k_crossprod_i8<false, 1>), grid, block, kF4Lds, s, d_X, nslabs, stages, d_tiles, rows, d_ans, ld, c0, nullptr, post);
    else hipLaunchKernelGGL((k_crossprod_i8<false, 2>), grid, block, kF4Lds, s, d_X, nslabs, stages, d_tiles, rows, d_ans, ld, c0, nullptr, post);
    MXA_HIP(hipGetLastError());
    return 0;
  }
  if (f4) {
    // (the diagnostic EXP instantiations of round 2 -- no unpack, no DMA, barrier only ..., all with wrong results -- were removed in round 5; what they measured: profiles/r02_mfma_f4_probe.txt)
    if (d_diag) hipLaunchKernelGGL(k_crossprod_f4<true>, grid, block, kF4Lds, s, d_X, nslabs, stages, d_tiles, rows, d_ans, ld, c0, d_diag, post);
    else hipLaunchKernelGGL(k_crossprod_f4<false>, grid, block, kF4Lds, s, d_X, nslabs, stages, d_tiles, rows, d_ans, ld, c0, d_diag, post);
  } else {
    static unsigned long long i0 = 0, i1 = 0;
    if (ensure_dyn_lds(reinterpret_cast<const void *>(&k_crossprod_i8<false>), kF4Lds, &i0) || ensure_dyn_lds(reinterpret_cast<const void *>(&k_crossprod_i8<true>), kF4Lds, &i1)) return 1;
    if (d_diag) hipLaunchKernelGGL(k_crossprod_i8<true>, grid, block, kF4Lds, s, d_X, nslabs, stages, d_tiles, rows, d_ans, ld, c0, d_diag, post);
    else hipLaunchKernelGGL(k_crossprod_i8<false>, grid, block, kF4Lds, s, d_X, nslabs, stages, d_tiles, rows, d_ans, ld, c0, d_diag, post);
  }
  MXA_HIP(hipGetLastError());
  return 0;
}

// X: device, tiled layout (rows padded to 256, K padded to 128 genotypes = nslabs slabs), zero padded.
// Columns [c_begin, c_end) of M = X X^T into d_ans (leading dimension ld; c_begin a multiple of the 256-row tile, c_end a multiple
// or the matrix end).  upper_only: only rows [0, c_end) are written -- everything above the panel's diagonal block and the block
// itself; rows >= c_end are left untouched.  The whole matrix is c_begin = 0, c_end = rows, ld = rows.
int crossprod_device(const uint8_t *d_X, long k, long rows, size_t pitch, double *d_ans, hipStream_t s, long c_begin, long c_end, bool upper_only,
                     long ld, bool f4, int post_kind, const XPost *post) {
  const int nb = (int)((rows + kXT - 1) / kXT);
  const int stages = (int)((k + kXStageK - 1) / kXStageK);
  const long nslabs = (long)(pitch / kXStageBytes);
  if (stages > nslabs) { set_error(4, "internal: crossproduct pitch too small"); return 1; }
  if (c_begin % kXT != 0 || c_begin < 0 || c_end > rows || c_begin >= c_end || ld < (upper_only ? c_end : rows)) { set_error(4, "crossproduct: bad column panel"); return 1; }
  const int t0 = (int)(c_begin / kXT), t1 = (int)((c_end + kXT - 1) / kXT);
  const bool whole = c_begin == 0 && c_end == rows;
  if (!whole && c_end % kXT != 0 && c_end != rows) { set_error(4, "crossproduct: panel end must be a multiple of %d or the matrix end", kXT); return 1; }
  // upper-triangular tiles (i <= j) that touch the panel: the direct image M[J rows, I cols] lands in the panel when i is a panel
  // column tile, the mirror image M[I rows, J cols] when j is
  std::vector<int4> tiles;
  for (int i = 0; i < nb; i++)
    for (int j = i; j < nb; j++) {
      int flags = 0;
      if (i >= t0 && i < t1 && (!upper_only || j < t1)) flags |= 1;      // rows of tile j >= i: on/below the diagonal
      if (j >= t0 && j < t1 && i != j) flags |= 2;                        // rows of tile i < j: above the diagonal
      if (flags) tiles.push_back(make_int4(i, j, flags, 0));
    }
  if (tiles.empty()) return 0;
  constexpr int gang_order = 1;   // (0: the 8 x 8 super-tiles of round 2, the A/B baseline)
  const bool xcd_lists = gang_order ? gang_order_tiles(tiles) : xcd_order_tiles(tiles, nb, 8);
  XBuf d_tiles, d_diag, d_gang;
  const size_t mid_cap = 7 * (tiles.size() + 64);      // >= 8 lists x slots per list x up to 7 meetings per tile
  if (d_gang.alloc(sizeof(int) * (32 + mid_cap))) return 1;
  if (d_tiles.alloc(tiles.size() * sizeof(int4))) return 1;
  MXA_HIP(hipMemcpyAsync(d_tiles.p, tiles.data(), tiles.size() * sizeof(int4), hipMemcpyHostToDevice, s));
  XEvent e0, e1;
  if (e0.create() || e1.create()) return 1;
  const bool diag_on = getenv("MXA_DIAG") != nullptr;
  if (diag_on && d_diag.alloc(16 * tiles.size())) return 1;
  MXA_HIP(hipEventRecord(e0.e, s));
  if (launch_tiles(f4, tiles.size(), s, d_X, nslabs, stages, (const int4 *)d_tiles.p, rows, d_ans, ld, c_begin, (unsigned long long *)d_diag.p, post ? post_kind : 0, post ? *post : XPost(),
                   xcd_lists ? (int *)d_gang.p : nullptr, mid_cap)) return 1;
  MXA_HIP(hipEventRecord(e1.e, s));
  MXA_HIP(hipStreamSynchronize(s));   // tiles vector / d_tiles lifetime
  if (diag_on) {   // diagnostic instantiation: in-kernel clock and cycles per stage
    std::vector<unsigned long long> hd(2 * tiles.size());
    MXA_HIP(hipMemcpy(hd.data(), d_diag.p, 16 * tiles.size(), hipMemcpyDeviceToHost));
    std::vector<double> ghz, cyc;
    for (size_t i = 0; i < tiles.size(); i++) if (tiles[i].z && hd[2 * i + 1]) { ghz.push_back((double)hd[2 * i] / (double)hd[2 * i + 1] * 0.1); cyc.push_back((double)hd[2 * i] / stages); }
    std::sort(ghz.begin(), ghz.end()); std::sort(cyc.begin(), cyc.end());
    if (!ghz.empty()) printf("MXA_DIAG %s: %zu tiles, in-kernel clock median %.3f GHz (min %.3f max %.3f); shader cycles per stage median %.0f (ideal %d)\n",
                             f4 ? "k_crossprod_f4" : "k_crossprod_i8", tiles.size(), ghz[ghz.size() / 2], ghz.front(), ghz.back(), cyc[cyc.size() / 2], f4 ? 1024 : 2048);
  }
  float ms = 0.f;
  MXA_HIP(hipEventElapsedTime(&ms, e0.e, e1.e));
  { std::lock_guard<std::mutex> lk(g_xprof_mutex); profile().launches += 1; profile().total_ms += ms; }
  return 0;
}

// Whole matrix for a HOST result (the plain reference ABI): the upper-triangular tiles are launched in chunks of tile rows i
// (all j >= i).  Tile (i, j) stores M[J rows, I cols] and M[I rows, J cols], so once every chunk up to tile row i1 has run, columns
// [0, 256*i1) of M are final: a helper thread copies each finished column slab to the host on its own non-blocking stream while
// the next chunk computes (at config 3 the 80 GB device-to-host copy is as long as the compute).
static int crossprod_to_host(const uint8_t *d_X, long k, long rows, size_t pitch, double *d_ans, double *h_ans, hipStream_t s, bool f4, int post_kind = 0,
                             const XPost *post = nullptr, const HostPrefault *pf = nullptr) {
  const int nb = (int)((rows + kXT - 1) / kXT);
  const int stages = (int)((k + kXStageK - 1) / kXStageK);
  const long nslabs = (long)(pitch / kXStageBytes);
  if (stages > nslabs) { set_error(4, "internal: crossproduct pitch too small"); return 1; }
  const char *slab_env = getenv("MXA_XPROD_SLAB_MB");                                        // tests use small slabs
  const long slab_bytes = (slab_env && atol(slab_env) > 0 ? atol(slab_env) : 1024L) << 20;
  const int rows_per_chunk = (int)std::max<long>(1, slab_bytes / (rows * 8 * kXT));         // ~1 GiB column slabs
  const int nchunks = (nb + rows_per_chunk - 1) / rows_per_chunk;
  constexpr int gang_order = 1;
  // one tile list per chunk of tile rows, each in XCD-aware order (super-tiles = the chunk's rows x 8 tile columns)
  std::vector<int4> tiles;
  std::vector<size_t> first((size_t)nchunks + 1, 0);
  std::vector<char> chunk_xcd((size_t)nchunks, 0);      // the chunk's list has the interleaved per-XCD form (gang-synchronised kernel)
  for (int c = 0; c < nchunks; c++) {
    first[(size_t)c] = tiles.size();
    std::vector<int4> part;
    for (int i = c * rows_per_chunk; i < std::min(nb, (c + 1) * rows_per_chunk); i++)
      for (int j = i; j < nb; j++) part.push_back(make_int4(i, j, i == j ? 1 : 3, 0));
    chunk_xcd[(size_t)c] = gang_order ? gang_order_tiles(part) : xcd_order_tiles(part, nb, std::min(rows_per_chunk, 8));
    tiles.insert(tiles.end(), part.begin(), part.end());
  }
  first[(size_t)nchunks] = tiles.size();
  XBuf d_tiles, d_gang;
  size_t mid_cap = 0;   // counters of the gangs' meetings inside a tile: sized for the longest slab list
  for (int c = 0; c < nchunks; c++) mid_cap = std::max(mid_cap, 7 * (first[(size_t)c + 1] - first[(size_t)c] + 64));
  if (d_tiles.alloc(tiles.size() * sizeof(int4)) || d_gang.alloc(sizeof(int) * (32 + mid_cap))) return 1;
  MXA_HIP(hipMemcpyAsync(d_tiles.p, tiles.data(), tiles.size() * sizeof(int4), hipMemcpyHostToDevice, s));
  std::vector<XEvent> ev((size_t)nchunks);
  int dev = 0;
  MXA_HIP(hipGetDevice(&dev));
  for (auto &e : ev) if (e.create(hipEventDisableTiming)) return 1;
  XEvent e0, e1;
  if (e0.create() || e1.create()) return 1;
  std::atomic<int> launched{0}, copy_err{0};
  std::atomic<bool> abort_copy{false};
  // a copy into pageable memory is staged by the runtime and bound by one host thread's memcpy (~17 GB/s measured): several
  // copier threads, each with its own stream and its own share of every slab, run those memcpys side by side
  constexpr int kCopiers = 4;
  XStream cs[kCopiers];
  for (auto &c : cs) if (c.create(hipStreamNonBlocking)) return 1;
  // where the host time of a call goes, per copier: waiting for a slab to be computed, and inside the copies; the slowest single copy with its
  // slab (a stall shows up there).  Printed under PRINT_LEVEL / print_details (debug_info) together with the prefault of the destination.
  struct CopierLog { double wait_s = 0, copy_s = 0, worst_s = 0; int worst_slab = -1; size_t bytes = 0; };
  CopierLog clog[kCopiers];
  const auto t_call = std::chrono::steady_clock::now();
  auto since = [&](std::chrono::steady_clock::time_point a) { return std::chrono::duration<double>(std::chrono::steady_clock::now() - a).count(); };
  auto copy_loop = [&](int t) {
    if (hipSetDevice(dev) != hipSuccess) { copy_err = 1; return; }
    for (int c = 0; c < nchunks; c++) {
      const auto tw = std::chrono::steady_clock::now();
      while (launched.load() <= c) { if (abort_copy.load()) return; std::this_thread::yield(); }
      if (hipEventSynchronize(ev[c].e) != hipSuccess) { copy_err = 1; return; }
      clog[t].wait_s += since(tw);
      const long col0 = (long)c * rows_per_chunk * kXT, col1 = std::min<long>(rows, (long)(c + 1) * rows_per_chunk * kXT);
      const long w = col1 - col0, a = col0 + w * t / kCopiers, b = col0 + w * (t + 1) / kCopiers;
      if (b <= a) continue;
      const size_t off = (size_t)a * rows, cnt = (size_t)(b - a) * rows;
      if (pf) { const auto tp = std::chrono::steady_clock::now(); pf->wait_for(h_ans + off + cnt); clog[t].wait_s += since(tp); }
      const auto tc = std::chrono::steady_clock::now();
      if (hipMemcpyAsync(h_ans + off, d_ans + off, cnt * sizeof(double), hipMemcpyDeviceToHost, cs[t].s) != hipSuccess || hipStreamSynchronize(cs[t].s) != hipSuccess) { copy_err = 1; return; }
      const double dt = since(tc);
      clog[t].copy_s += dt; clog[t].bytes += cnt * sizeof(double);
      if (dt > clog[t].worst_s) { clog[t].worst_s = dt; clog[t].worst_slab = c; }
    }
  };
  std::vector<std::thread> copiers;
  for (int t = 0; t < kCopiers; t++) copiers.emplace_back(copy_loop, t);
  int rc = 0;
  if (hipEventRecord(e0.e, s) != hipSuccess) rc = 1;
  for (int c = 0; c < nchunks && !rc; c++) {
    const size_t cnt = first[(size_t)c + 1] - first[(size_t)c];
    if (launch_tiles(f4, cnt, s, d_X, nslabs, stages, (const int4 *)d_tiles.p + first[(size_t)c], rows, d_ans, rows, 0L, nullptr, post ? post_kind : 0, post ? *post : XPost(), chunk_xcd[(size_t)c] ? (int *)d_gang.p : nullptr, mid_cap) || hipEventRecord(ev[c].e, s) != hipSuccess) { rc = 1; break; }
    launched.store(c + 1);
  }
  if (rc) abort_copy = true;
  if (!rc && hipEventRecord(e1.e, s) != hipSuccess) rc = 1;
  const double t_launched = since(t_call);
  for (auto &t : copiers) t.join();
  if (hipStreamSynchronize(s) != hipSuccess) rc = 1;
  for (int t = 0; t < kCopiers; t++)
    debug_info("crossproduct host result: copier %d waited %.3f s for slabs, copied %.2f GB in %.3f s (%.1f GB/s), slowest single copy %.3f s (slab %d of %d)", t, clog[t].wait_s,
               clog[t].bytes * 1e-9, clog[t].copy_s, clog[t].copy_s > 0 ? clog[t].bytes * 1e-9 / clog[t].copy_s : 0.0, clog[t].worst_s, clog[t].worst_slab, nchunks);
  debug_info("crossproduct host result: %d slab launches enqueued after %.3f s, all copies done after %.3f s", nchunks, t_launched, since(t_call));
  if (!rc && !copy_err.load()) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, e0.e, e1.e) == hipSuccess) { profile().launches += 1; profile().total_ms += ms; }
  }
  if (rc || copy_err.load()) { set_error(13, "snp_multiply_gpu: pipelined device-to-host copy of the result failed"); return 1; }
  return 0;
}

// Host result WITHOUT a device copy of the whole matrix (round 4).  crossprod_to_host above needs the full n x n device buffer (80 GB at config 3):
// hipMalloc of such a buffer takes anything from nothing to 4.6 s on this pool (the phase clock of crossprod_any, profiles/r04_crossprod_host_*.txt:
// that -- not a copy scheme -- was round 3's unexplained "one call in 24 takes 4-5 s"), and the result could never exceed HBM.  Here the matrix is
// produced column slab by column slab into a RING of three ~1 GiB device buffers: slab [c0, c1) is the column panel of mxa_snp_multiply_panel (every
// tile (i, j >= i) that touches it: its direct image when i lies in the panel, its mirror image when j does), computed by the same kernels, and leaves
// through the four copier threads while the next slab is computed.  Every off-diagonal tile is computed twice (once per image): twice the arithmetic of
// the triangular launch -- taken only where the call is bound by the copy anyway (the caller compares the two estimates).
static int crossprod_to_host_ring(const uint8_t *d_X, long k, long rows, size_t pitch, double *h_ans, hipStream_t s, bool f4, int post_kind, const XPost *post,
                                  const std::function<void()> &all_allocated, const HostPrefault *pf) {
  const int nb = (int)((rows + kXT - 1) / kXT);
  const int stages = (int)((k + kXStageK - 1) / kXStageK);
  const long nslabs = (long)(pitch / kXStageBytes);
  if (stages > nslabs) { set_error(4, "internal: crossproduct pitch too small"); return 1; }
  const char *slab_env = getenv("MXA_XPROD_SLAB_MB");                                        // tests use small slabs
  const long slab_bytes = (slab_env && atol(slab_env) > 0 ? atol(slab_env) : 1024L) << 20;
  const int tcols = (int)std::max<long>(1, slab_bytes / (rows * 8 * kXT));                   // tile columns per slab
  const int nchunks = (nb + tcols - 1) / tcols;
  constexpr int kRing = 3, kCopiers = 4;
  const size_t slot_elems = (size_t)rows * (size_t)std::min<long>(rows, (long)tcols * kXT);
  XBuf ring[kRing], d_tiles, d_gang;
  for (auto &r : ring) if (r.alloc(slot_elems * sizeof(double))) return 1;
  constexpr int gang_order = 1;
  // tile lists of all slabs, one after the other (uploaded once)
  std::vector<int4> tiles;
  std::vector<size_t> first((size_t)nchunks + 1, 0);
  std::vector<char> chunk_xcd((size_t)nchunks, 0);
  for (int c = 0; c < nchunks; c++) {
    first[(size_t)c] = tiles.size();
    const int t0 = c * tcols, t1 = std::min(nb, t0 + tcols);
    std::vector<int4> part;
    for (int i = 0; i < nb; i++)
      for (int j = i; j < nb; j++) {
        int flags = 0;
        if (i >= t0 && i < t1) flags |= 1;                   // direct image M[J rows, I cols]: columns of tile i
        if (j >= t0 && j < t1 && i != j) flags |= 2;         // mirror image M[I rows, J cols]: columns of tile j
        if (flags) part.push_back(make_int4(i, j, flags, 0));
      }
    chunk_xcd[(size_t)c] = gang_order ? gang_order_tiles(part) : xcd_order_tiles(part, nb, 8);
    tiles.insert(tiles.end(), part.begin(), part.end());
  }
  first[(size_t)nchunks] = tiles.size();
  size_t mid_cap = 0;   // counters of the gangs' meetings inside a tile: sized for the longest slab list
  for (int c = 0; c < nchunks; c++) mid_cap = std::max(mid_cap, 7 * (first[(size_t)c + 1] - first[(size_t)c] + 64));
  if (d_tiles.alloc(tiles.size() * sizeof(int4)) || d_gang.alloc(sizeof(int) * (32 + mid_cap))) return 1;
  MXA_HIP(hipMemcpyAsync(d_tiles.p, tiles.data(), tiles.size() * sizeof(int4), hipMemcpyHostToDevice, s));
  std::vector<XEvent> ev((size_t)nchunks);
  int dev = 0;
  MXA_HIP(hipGetDevice(&dev));
  for (auto &e : ev) if (e.create(hipEventDisableTiming)) return 1;
  XEvent e0, e1;
  if (e0.create() || e1.create()) return 1;
  std::atomic<int> launched{0}, copy_err{0};
  std::atomic<bool> abort_copy{false};
  std::vector<std::atomic<int>> copied((size_t)nchunks);
  for (auto &c : copied) c.store(0);
  XStream cs[kCopiers];
  for (auto &c : cs) if (c.create(hipStreamNonBlocking)) return 1;
  struct CopierLog { double wait_s = 0, copy_s = 0, worst_s = 0; int worst_slab = -1; size_t bytes = 0; };
  CopierLog clog[kCopiers];
  const auto t_call = std::chrono::steady_clock::now();
  auto since = [&](std::chrono::steady_clock::time_point a) { return std::chrono::duration<double>(std::chrono::steady_clock::now() - a).count(); };
  auto copy_loop = [&](int t) {
    if (hipSetDevice(dev) != hipSuccess) { copy_err = 1; abort_copy = true; return; }
    for (int c = 0; c < nchunks; c++) {
      const auto tw = std::chrono::steady_clock::now();
      while (launched.load() <= c) { if (abort_copy.load()) return; std::this_thread::yield(); }
      if (hipEventSynchronize(ev[c].e) != hipSuccess) { copy_err = 1; abort_copy = true; return; }
      clog[t].wait_s += since(tw);
      const long col0 = (long)c * tcols * kXT, col1 = std::min<long>(rows, (long)(c + 1) * tcols * kXT);
      const long w = col1 - col0, a = w * t / kCopiers, b = w * (t + 1) / kCopiers;   // columns of the slab, relative
      if (b > a) {
        const double *src = (const double *)ring[c % kRing].p + (size_t)a * rows;
        const size_t cnt = (size_t)(b - a) * rows;
        if (pf) { const auto tp = std::chrono::steady_clock::now(); pf->wait_for(h_ans + (size_t)(col0 + a) * rows + cnt); clog[t].wait_s += since(tp); }   // the destination pages exist: no faults inside the copy
        const auto tc = std::chrono::steady_clock::now();
        if (hipMemcpyAsync(h_ans + (size_t)(col0 + a) * rows, src, cnt * sizeof(double), hipMemcpyDeviceToHost, cs[t].s) != hipSuccess || hipStreamSynchronize(cs[t].s) != hipSuccess) { copy_err = 1; abort_copy = true; return; }
        const double dt = since(tc);
        clog[t].copy_s += dt; clog[t].bytes += cnt * sizeof(double);
        if (dt > clog[t].worst_s) { clog[t].worst_s = dt; clog[t].worst_slab = c; }
      }
      copied[(size_t)c].fetch_add(1);
    }
  };
  all_allocated();   // every device buffer, stream and event of this call exists: the background population of the destination pages may start (mxa_hostmem.h)
  std::vector<std::thread> copiers;
  for (int t = 0; t < kCopiers; t++) copiers.emplace_back(copy_loop, t);
  int rc = 0;
  double t_wait_slot = 0.0;
  if (hipEventRecord(e0.e, s) != hipSuccess) rc = 1;
  for (int c = 0; c < nchunks && !rc; c++) {
    if (c >= kRing) {   // the slot is free once all copiers have taken slab c - kRing out of it
      const auto tw = std::chrono::steady_clock::now();
      while (copied[(size_t)(c - kRing)].load() < kCopiers) { if (abort_copy.load()) { rc = 1; break; } std::this_thread::yield(); }
      t_wait_slot += since(tw);
      if (rc) break;
    }
    const size_t cnt = first[(size_t)c + 1] - first[(size_t)c];
    const long col0 = (long)c * tcols * kXT;
    if (launch_tiles(f4, cnt, s, d_X, nslabs, stages, (const int4 *)d_tiles.p + first[(size_t)c], rows, (double *)ring[c % kRing].p, rows, col0, nullptr, post ? post_kind : 0, post ? *post : XPost(),
                     chunk_xcd[(size_t)c] ? (int *)d_gang.p : nullptr, mid_cap) || hipEventRecord(ev[c].e, s) != hipSuccess) { rc = 1; break; }
    launched.store(c + 1);
  }
  if (rc) abort_copy = true;
  if (!rc && hipEventRecord(e1.e, s) != hipSuccess) rc = 1;
  const double t_launched = since(t_call);
  for (auto &t : copiers) t.join();
  if (hipStreamSynchronize(s) != hipSuccess) rc = 1;
  for (int t = 0; t < kCopiers; t++)
    debug_info("crossproduct host result (ring of %d slabs): copier %d waited %.3f s for slabs, copied %.2f GB in %.3f s (%.1f GB/s), slowest single copy %.3f s (slab %d of %d)", kRing, t, clog[t].wait_s,
               clog[t].bytes * 1e-9, clog[t].copy_s, clog[t].copy_s > 0 ? clog[t].bytes * 1e-9 / clog[t].copy_s : 0.0, clog[t].worst_s, clog[t].worst_slab, nchunks);
  debug_info("crossproduct host result (ring): %d slab launches enqueued after %.3f s (%.3f s of it waiting for a free slot), all copies done after %.3f s", nchunks, t_launched, t_wait_slot, since(t_call));
  if (!rc && !copy_err.load()) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, e0.e, e1.e) == hipSuccess) { std::lock_guard<std::mutex> lk(g_xprof_mutex); profile().launches += 1; profile().total_ms += ms; }
  }
  if (rc || copy_err.load()) { set_error(13, "snp_multiply_gpu: pipelined device-to-host copy of the result failed"); return 1; }
  return 0;
}

// ---- GRM / LD post-processing on the device (reference: host BLAS in src/bindings/Julia/crossproduct.jl:83-152, maths docs/grm.md)
// column sums of the symmetric n x n matrix, fixed-order tree per column
__global__ void __launch_bounds__(256) k_sym_colsum(const double *__restrict__ M, long n, double *__restrict__ cs) {
  const long j = blockIdx.x;
  double s = 0.0;
  for (long i = threadIdx.x; i < n; i += 256) s += M[(size_t)j * n + i];
  __shared__ double sh[256];
  sh[threadIdx.x] = s;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) { if ((int)threadIdx.x < w) sh[threadIdx.x] += sh[threadIdx.x + w]; __syncthreads(); }
  if (threadIdx.x == 0) cs[j] = sh[0];
}
// out[0] = sum_i v[i] * (w ? (1 - w[i]) * 2 : 1)   (single block, fixed order)
__global__ void __launch_bounds__(1024) k_vec_reduce(const double *__restrict__ v, long n, int mode, double *__restrict__ out) {
  double s = 0.0;
  for (long i = threadIdx.x; i < n; i += 1024) s += mode ? 2.0 * v[i] * (1.0 - v[i]) : v[i];
  __shared__ double sh[1024];
  sh[threadIdx.x] = s;
  __syncthreads();
  for (int w = 512; w > 0; w >>= 1) { if ((int)threadIdx.x < w) sh[threadIdx.x] += sh[threadIdx.x + w]; __syncthreads(); }
  if (threadIdx.x == 0) out[0] = sh[0];
}
// G = (M - cs 1^T / n - 1 cs^T / n + total / n^2) / c          (crossproduct.jl:96-107)
__global__ void __launch_bounds__(256) k_grm_update(double *__restrict__ M, long n, const double *__restrict__ cs, const double *__restrict__ total,
                                                    const double *__restrict__ c, int do_scale) {
  const long j = blockIdx.x;                               // column: gridDim.x may exceed 65535, gridDim.y may not
  const long i = (long)blockIdx.y * 256 + threadIdx.x;
  if (i >= n) return;
  const double inv_n = 1.0 / (double)n;
  M[(size_t)j * n + i] = grm_map(M[(size_t)j * n + i], cs[i], cs[j], inv_n, total[0] / ((double)n * (double)n), do_scale ? 1.0 / c[0] : 1.0, do_scale);
}
// LD: M <- M - 4 * indiv * f f^T ; sigma = sqrt(diag M) ; M <- M / sigma sigma^T      (crossproduct.jl:139-149)
__global__ void __launch_bounds__(256) k_ld_center(double *__restrict__ M, long n, const double *__restrict__ f, double four_indiv) {
  const long j = blockIdx.x;
  const long i = (long)blockIdx.y * 256 + threadIdx.x;
  if (i >= n) return;
  M[(size_t)j * n + i] = ld_center_map(M[(size_t)j * n + i], f[i], f[j], four_indiv);
}
__global__ void __launch_bounds__(256) k_diag_sqrt(const double *__restrict__ M, long n, double *__restrict__ sigma) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i < n) sigma[i] = 1.0 / sqrt(M[(size_t)i * n + i]);   // reciprocal: ld_scale_map multiplies
}
__global__ void __launch_bounds__(256) k_ld_scale(double *__restrict__ M, long n, const double *__restrict__ sigma) {
  const long j = blockIdx.x;
  const long i = (long)blockIdx.y * 256 + threadIdx.x;
  if (i >= n) return;
  M[(size_t)j * n + i] = ld_scale_map(M[(size_t)j * n + i], sigma[i], sigma[j]);
}

// ---- what the fused map needs, from the staged 2-bit matrix X (tiled layout, values 0..3) instead of from the 8 n^2-byte result ----------------
// t[s] = sum over all rows of x[r][s] (int32: <= 3 rows).  One block per slab of 128 genotypes walks all row tiles; thread tid reads dword
// it * 256 + tid of every 8-KiB tile (lane-linear), i.e. always dword `tid & 7` of a row piece: 16 fixed genotype columns per thread.  SWAR byte
// counters (4 fields per register), flushed to 16 int32 counters before they can overflow; fixed-order reduction over the 32 threads of a column group.
__global__ void __launch_bounds__(256) k_x_colsum(const uint8_t *__restrict__ X, long nslabs, long ntiles, int *__restrict__ t) {
  const long slab = blockIdx.x;
  const int tid = threadIdx.x;
  uint32_t a0 = 0, a1 = 0, a2 = 0, a3 = 0;
  int cnt[16];
#pragma unroll
  for (int f = 0; f < 16; f++) cnt[f] = 0;
  auto flush = [&]() {
#pragma unroll
    for (int q = 0; q < 4; q++) {
      cnt[4 * q + 0] += (int)((a0 >> (8 * q)) & 255u); cnt[4 * q + 1] += (int)((a1 >> (8 * q)) & 255u);
      cnt[4 * q + 2] += (int)((a2 >> (8 * q)) & 255u); cnt[4 * q + 3] += (int)((a3 >> (8 * q)) & 255u);
    }
    a0 = a1 = a2 = a3 = 0;
  };
  int pend = 0;
  for (long rt = 0; rt < ntiles; rt++) {
    const uint32_t *tile = reinterpret_cast<const uint32_t *>(X + ((size_t)rt * nslabs + (size_t)slab) * kTileBytes);
#pragma unroll
    for (int it = 0; it < 8; it++) {
      const uint32_t w = tile[it * 256 + tid];
      a0 += w & 0x03030303u; a1 += (w >> 2) & 0x03030303u; a2 += (w >> 4) & 0x03030303u; a3 += (w >> 6) & 0x03030303u;
    }
    pend += 8;                                   // every byte counter grew by at most 3 * 8
    if (pend + 8 > 85) { flush(); pend = 0; }    // 85 * 3 = 255
  }
  flush();
  __shared__ int sh[256][17];
#pragma unroll
  for (int f = 0; f < 16; f++) sh[tid][f] = cnt[f];
  __syncthreads();
  if (tid < 128) {
    const int part = tid >> 4, f = tid & 15;     // genotype part * 16 + f of the slab = field f of dword `part` of the row piece
    int sum = 0;
    for (int g = 0; g < 32; g++) sum += sh[g * 8 + part][f];
    t[slab * 128 + tid] = sum;
  }
}
// per row r: cs[r] += sum_s x[r][s] * t[s]  (WANT_CS;  = column sum r of M = X X^T)  and / or  dg[r] += sum_s x[r][s]^2  (WANT_DG; = M[r][r]).
// Grid (row tiles, K chunks); thread = row of the tile, reading its 32-byte piece of every slab of the chunk (a wave reads 2 KiB contiguous);
// the t values of a slab are broadcast from LDS.  Exact integers: 32 bits within a dword of 16 fields (rows < 29.8 M: kXFusedMaxRows), 64 bits beyond, 64-bit atomics
// across the chunks (integer addition: order-independent).
template <bool WANT_CS, bool WANT_DG>
__global__ void __launch_bounds__(256) k_x_rowstats(const uint8_t *__restrict__ X, long nslabs, long slabs_per_chunk, const int *__restrict__ t,
                                                    unsigned long long *__restrict__ cs, unsigned long long *__restrict__ dg) {
  const long rt = blockIdx.x;
  const long s0 = (long)blockIdx.y * slabs_per_chunk, s1 = min(nslabs, s0 + slabs_per_chunk);
  const int tid = threadIdx.x;
  __shared__ int tsh[128];
  unsigned long long acc_cs = 0, acc_dg = 0;
  for (long sl = s0; sl < s1; sl++) {
    if (WANT_CS) {
      __syncthreads();
      if (tid < 128) tsh[tid] = t[sl * 128 + tid];
      __syncthreads();
    }
    const uint4 *pp = reinterpret_cast<const uint4 *>(X + ((size_t)rt * nslabs + (size_t)sl) * kTileBytes + (size_t)tid * kSlabBytes);
    const uint4 q0 = pp[0], q1 = pp[1];
    const uint32_t w[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
    unsigned long long part_cs = 0;
    uint32_t part_dg = 0;
#pragma unroll
    for (int d = 0; d < 8; d++) {
      if (WANT_CS) {   // 32 bits hold the 16 fields of one dword (16 * 3 * 3 rows < 2^32 for rows < 29.8 M, guarded by the caller); across dwords 64 bits
        uint32_t pd = 0;
#pragma unroll
        for (int f = 0; f < 16; f++) pd += ((w[d] >> (2 * f)) & 3u) * (uint32_t)tsh[d * 16 + f];
        part_cs += pd;
      }
      if (WANT_DG) {   // x^2 = 1, 4, 9 for x = 1, 2, 3: three bit counts
        const uint32_t L = w[d] & 0x55555555u, H = (w[d] >> 1) & 0x55555555u;
        part_dg += (uint32_t)__popc(L & ~H) + 4u * (uint32_t)__popc(H & ~L) + 9u * (uint32_t)__popc(H & L);
      }
    }
    acc_cs += part_cs; acc_dg += part_dg;
  }
  const long r = rt * kTileRows + tid;
  if (WANT_CS && acc_cs) atomicAdd(cs + r, acc_cs);
  if (WANT_DG && acc_dg) atomicAdd(dg + r, acc_dg);
}
// u64 -> double (exact below 2^53), or the reciprocal LD sigma: 1 / sqrt(M_ii - 4 indiv f_i^2) exactly as k_ld_center + k_diag_sqrt form it
__global__ void __launch_bounds__(256) k_x_finish_stats(const unsigned long long *__restrict__ in, long n, const double *__restrict__ f, double four_indiv, double *__restrict__ out) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const double v = (double)in[i];
  out[i] = f ? 1.0 / sqrt(ld_center_map(v, f[i], f[i], four_indiv)) : v;
}

// post: 0 none, 1 GRM (do_scale as given, f = allele frequencies of length k), 2 LD (f of length rows, k = number of individuals)
static int postprocess_device(double *d_M, long rows, long k, int post, int do_scale, const double *d_f, hipStream_t s) {
  if (post == 0) return 0;
  XBuf tmp_buf;
  if (tmp_buf.alloc(sizeof(double) * (size_t)(rows + 4))) return 1;
  double *tmp = (double *)tmp_buf.p;
  dim3 g2((unsigned)rows, (unsigned)((rows + 255) / 256));   // x = column (unbounded), y = row chunk (<= 65535)
  if (post == 1) {
    hipLaunchKernelGGL(k_sym_colsum, dim3((unsigned)rows), dim3(256), 0, s, d_M, rows, tmp);
    hipLaunchKernelGGL(k_vec_reduce, dim3(1), dim3(1024), 0, s, tmp, rows, 0, tmp + rows);
    if (do_scale) hipLaunchKernelGGL(k_vec_reduce, dim3(1), dim3(1024), 0, s, d_f, k, 1, tmp + rows + 1);
    hipLaunchKernelGGL(k_grm_update, g2, dim3(256), 0, s, d_M, rows, tmp, tmp + rows, tmp + rows + 1, do_scale);
  } else {
    hipLaunchKernelGGL(k_ld_center, g2, dim3(256), 0, s, d_M, rows, d_f, 4.0 * (double)k);
    hipLaunchKernelGGL(k_diag_sqrt, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, s, d_M, rows, tmp);
    hipLaunchKernelGGL(k_ld_scale, g2, dim3(256), 0, s, d_M, rows, tmp);
  }
  MXA_HIP(hipGetLastError());
  MXA_HIP(hipStreamSynchronize(s));
  return 0;
}

static int crossprod_any(const unsigned char *snp_matrix, long k, long rows, double *ans, bool is_plink, int post = 0, int do_scale = 0,
                         const double *freq = nullptr, long c_begin = 0, long c_end = -1, bool upper_only = false, long ld = -1, int device = -1) {
  if (c_end < 0) c_end = rows;
  if (ld < 0) ld = rows;
  if (!snp_matrix || !ans || k <= 0 || rows <= 0) { set_error(1, "snp_multiply_gpu: bad arguments"); return 1; }
  if (device >= 0) MXA_HIP(hipSetDevice(device));
  else if (select_device() < 0) return 1;   // HIP_DEVICE / CUDA_DEVICE with the range check; GPU-only
  const long row_bytes = (k + 3) / 4;
  const long rows_pad = (rows + kXT - 1) / kXT * kXT;
  const long nslabs = (k + kXStageK - 1) / kXStageK;
  const size_t pitch = (size_t)nslabs * kXStageBytes;
  const bool in_dev = ptr_location(snp_matrix, nullptr) == 1, out_dev = ptr_location(ans, nullptr) == 1;
  if (c_begin < 0 || c_begin >= c_end || c_end > rows || c_begin % kXT != 0 || (c_end % kXT != 0 && c_end != rows) || ld < (upper_only ? c_end : rows)) {
    set_error(1, "crossproduct panel: need 0 <= col_begin < col_end <= n, col_begin %% %d == 0, col_end %% %d == 0 or col_end == n, ld >= rows written", kXT, kXT);
    return 1;
  }
  const size_t xbytes = (size_t)rows_pad * pitch, abytes = (size_t)ld * (size_t)(c_end - c_begin) * sizeof(double);
  size_t free_b = 0, total_b = 0;
  MXA_HIP(hipMemGetInfo(&free_b, &total_b));
  // a whole-matrix host result can leave through a ring of three ~1 GiB slabs (crossprod_to_host_ring): the n x n device copy is then not needed
  const bool ring_ok = !out_dev && c_begin == 0 && c_end == rows && !upper_only && ld == rows && !getenv("MXA_XPROD_NO_PIPELINE");
  const size_t out_need = out_dev ? 0 : (ring_ok ? std::min<size_t>(abytes, (size_t)3400 << 20) : abytes);
  const size_t need = xbytes + out_need + (in_dev ? 0 : std::min<size_t>((size_t)rows * row_bytes, (size_t)256 << 20));
  if (need > free_b) { set_error(12, "snp_multiply_gpu: not enough device memory: required %zu GB, free %zu GB", need >> 30, free_b >> 30); return 1; }
  // a host result in fresh memory (crossproduct.jl:56 `M = zeros(...)`): its pages are populated in the background while the tiles are computed, so that
  // the copies do not pay the first-touch faults (mxa_hostmem.h).  Joined when this returns.
  // Started only AFTER the operand has been staged and EVERY device buffer of the call has been allocated: twelve threads inside madvise slow a concurrent
  // hipMalloc (12.5 GB: 1.2-1.3 s instead of < 0.06 s; the 3 GiB ring: 1.7 s) and the staged pageable upload (1.55 s instead of 0.24 s) by far more than
  // the head start is worth (profiles/r04_crossprod_host_abi_c3.txt).
  HostPrefault prefault;
  struct PrefaultReport {
    HostPrefault &p;
    ~PrefaultReport() {
      p.join();
      if (p.threads()) debug_info("host result: %.2f GB of destination pages populated in the background by %d threads in %.3f s%s", p.populated() * 1e-9, p.threads(), p.seconds(),
                                  p.unsupported() ? " (MADV_POPULATE_WRITE not supported by this kernel: first-touch faults stay in the copies)" : "");
    }
  } prefault_report{prefault};
  // phase clock of a call with a host operand (debug_info under PRINT_LEVEL): where the wall time of the plain ABI goes
  struct PhaseClock {
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now(), last = t0;
    bool on;
    explicit PhaseClock(bool o) : on(o) {}
    void mark(const char *what) {
      if (!on) return;
      const auto now = std::chrono::steady_clock::now();
      debug_info("crossproduct call: %-34s %.3f s (at %.3f s)", what, std::chrono::duration<double>(now - last).count(), std::chrono::duration<double>(now - t0).count());
      last = now;
    }
  } clk(true);   // (round 5: for device operands too -- the operand is re-tiled into a buffer allocated per call, and hipMalloc of 12.5 GB takes 0-1.3 s on this pool)
  XStream st;
  if (st.create(hipStreamDefault)) return 1;   // blocking: ordered against the caller's default-stream work
  hipStream_t s = st.s;
  XBuf d_X, bounce, d_out, d_flag, f_tmp;
  // the n x n device buffer of a host result that does not take the slab ring: checked against the free memory first, so that what does not fit is reported
  // like the reference's pre-flight (cuda_utils.cu:162-185) instead of as a raw hipMalloc failure
  auto alloc_result = [&](XBuf &b, size_t bytes) -> int {
    size_t fb = 0, tb = 0;
    if (hipMemGetInfo(&fb, &tb) == hipSuccess && bytes > fb) {
      set_error(12, "Not enough device memory available. Required %zu GB, free %zu GB, total on device %zu GB", bytes >> 30, fb >> 30, tb >> 30);
      return 1;
    }
    (void)hipGetLastError();
    return b.alloc(bytes);
  };
  if (d_X.alloc(xbytes) || d_flag.alloc(sizeof(int))) return 1;
  clk.mark("operand buffer allocated");
  MXA_HIP(hipMemsetAsync(d_X.p, 0, xbytes, s));
  MXA_HIP(hipMemsetAsync(d_flag.p, 0, sizeof(int), s));
  if (in_dev) {
    const long total = rows * ((row_bytes + 3) / 4);
    hipLaunchKernelGGL(k_xstage, dim3((unsigned)std::min<long>((total + 255) / 256, 8192)), dim3(256), 0, s, snp_matrix, (size_t)row_bytes, row_bytes, rows, (uint8_t *)d_X.p, nslabs, 0L,
                       is_plink ? 1 : 0, (int *)d_flag.p);
    MXA_HIP(hipGetLastError());
  } else {
    long chunk_rows = std::max<long>(1, (long)(((size_t)256 << 20) / (size_t)row_bytes));
    chunk_rows = std::min(chunk_rows, rows);
    if (bounce.alloc((size_t)chunk_rows * row_bytes)) return 1;
    for (long r0 = 0; r0 < rows; r0 += chunk_rows) {
      const long nr = std::min(chunk_rows, rows - r0);
      MXA_HIP(hipMemcpyAsync(bounce.p, snp_matrix + (size_t)r0 * row_bytes, (size_t)nr * row_bytes, hipMemcpyHostToDevice, s));
      const long total = nr * ((row_bytes + 3) / 4);
      hipLaunchKernelGGL(k_xstage, dim3((unsigned)std::min<long>((total + 255) / 256, 8192)), dim3(256), 0, s, (const uint8_t *)bounce.p, (size_t)row_bytes, row_bytes, nr, (uint8_t *)d_X.p,
                         nslabs, r0, is_plink ? 1 : 0, (int *)d_flag.p);
      MXA_HIP(hipGetLastError());
      MXA_HIP(hipStreamSynchronize(s));
    }
  }
  clk.mark("operand staged (upload + k_xstage)");
  bool prefault_started = false;
  auto start_prefault = [&]() { if (!out_dev && !prefault_started) { prefault_started = true; prefault.start(ans, abytes); } };
  // engine: FP4 while the fp32 accumulator is provably exact (sum z z' < 2^24), int8 beyond (MXA_XPROD_ENGINE=i8 / f4 forces one, for A/B runs)
  int has3 = 1;
  MXA_HIP(hipMemcpyAsync(&has3, d_flag.p, sizeof(int), hipMemcpyDeviceToHost, s));
  MXA_HIP(hipStreamSynchronize(s));
  bool f4 = has3 ? 9 * k < (1L << 24) : 4 * k < (1L << 24);
  if (const char *e = getenv("MXA_XPROD_ENGINE")) { if (!strcmp(e, "i8")) f4 = false; }
  double *d_ans = ans;
  const bool whole0 = c_begin == 0 && c_end == rows && !upper_only;
  const char *e_fused0 = getenv("MXA_XPROD_FUSED_POST");
  const bool fused_possible = (!e_fused0 || atoi(e_fused0) != 0) && whole0;
  // host result of the whole matrix: ring of column slabs (no n x n device buffer) where the call is bound by the download anyway -- the ring computes
  // every off-diagonal tile twice.  Estimates: triangular arithmetic at the measured tile rate against the download at ~55 GB/s of four copiers.
  // MXA_XPROD_HOST_RING: 0 never, 1 by this estimate (default), 2 always (tests).
  bool use_ring = false;
  if (!out_dev && whole0 && ld == rows && !getenv("MXA_XPROD_NO_PIPELINE")) {
    const char *e_ring = getenv("MXA_XPROD_HOST_RING");
    const int ring_mode = e_ring ? atoi(e_ring) : 1;
    const double nbt = (double)((rows + kXT - 1) / kXT), tri_ms = (double)nslabs * (f4 ? 0.66e-3 : 1.0e-3) * nbt * (nbt + 1.0) / 2.0 / 256.0, copy_ms = (double)abytes / 55e9 * 1e3;
    use_ring = ring_mode >= 2 || (ring_mode == 1 && abytes >= ((size_t)4 << 30) && 2.0 * tri_ms <= 1.15 * copy_ms);
    if (!use_ring && ring_mode >= 1 && (!post || fused_possible)) {   // a result that does not fit the free device memory can only leave through the ring
      size_t fb = 0, tb = 0;
      if (hipMemGetInfo(&fb, &tb) == hipSuccess && abytes + ((size_t)1 << 30) > fb) use_ring = true;
    }
  }
  if (!out_dev && !use_ring) {
    if (alloc_result(d_out, abytes)) return 1;
    d_ans = (double *)d_out.p;
    if (upper_only) MXA_HIP(hipMemsetAsync(d_ans, 0, abytes, s));   // the untouched part travels back as zeros
    clk.mark("device result buffer allocated");
    start_prefault();
  } else if (!out_dev) d_ans = nullptr;
  const bool whole = c_begin == 0 && c_end == rows && !upper_only;
  // GRM / LD: the element-wise map is fused into the crossproduct epilogue (whole matrix; MXA_XPROD_FUSED_POST=0 keeps the three extra passes over
  // the result).  What the map needs comes from the staged 2-bit matrix: ~2 passes over rows * k / 4 bytes instead of 3 over 8 * rows^2.
  const char *e_fused = getenv("MXA_XPROD_FUSED_POST");   // read per call (tests compare both paths bit for bit)
  const bool fused_on = !e_fused || atoi(e_fused) != 0;
  const double *d_f = freq;
  if (post && freq && ptr_location(freq, nullptr) != 1) {
    const long flen = post == 1 ? k : rows;
    if (f_tmp.alloc(sizeof(double) * flen)) return 1;
    MXA_HIP(hipMemcpyAsync(f_tmp.p, freq, sizeof(double) * flen, hipMemcpyHostToDevice, s));
    d_f = (const double *)f_tmp.p;
  }
  XPost xp;
  int post_kind = 0;
  XBuf st_t, st_raw, st_out;
  constexpr long kXFusedMaxRows = 29000000L;   // k_x_rowstats: 16 * 3 * (3 rows) must fit 32 bits; beyond that the three-pass post-processing runs
  if (post && fused_on && whole && rows < kXFusedMaxRows) {
    const long ntiles = rows_pad / kXT;
    if (st_t.alloc(sizeof(int) * (size_t)nslabs * 128) || st_raw.alloc(sizeof(unsigned long long) * (size_t)rows_pad) || st_out.alloc(sizeof(double) * (size_t)(rows_pad + 4))) return 1;
    MXA_HIP(hipMemsetAsync(st_raw.p, 0, sizeof(unsigned long long) * (size_t)rows_pad, s));
    unsigned long long *raw = (unsigned long long *)st_raw.p;
    double *out = (double *)st_out.p;
    const long chunks = std::max<long>(1, std::min<long>(nslabs, (1024 + ntiles - 1) / ntiles));   // >= ~1024 blocks
    const long spc = (nslabs + chunks - 1) / chunks;
    const dim3 g_rows((unsigned)ntiles, (unsigned)((nslabs + spc - 1) / spc));
    const unsigned g_fin = (unsigned)((rows + 255) / 256);
    if (post == 1) {
      hipLaunchKernelGGL(k_x_colsum, dim3((unsigned)nslabs), dim3(256), 0, s, (const uint8_t *)d_X.p, nslabs, ntiles, (int *)st_t.p);
      hipLaunchKernelGGL((k_x_rowstats<true, false>), g_rows, dim3(256), 0, s, (const uint8_t *)d_X.p, nslabs, spc, (const int *)st_t.p, raw, (unsigned long long *)nullptr);
      hipLaunchKernelGGL(k_x_finish_stats, dim3(g_fin), dim3(256), 0, s, raw, rows, (const double *)nullptr, 0.0, out);
      hipLaunchKernelGGL(k_vec_reduce, dim3(1), dim3(1024), 0, s, out, rows, 0, out + rows_pad);
      if (do_scale) hipLaunchKernelGGL(k_vec_reduce, dim3(1), dim3(1024), 0, s, d_f, k, 1, out + rows_pad + 1);
      xp.u = out; xp.scal = out + rows_pad; xp.a = 1.0 / (double)rows; xp.do_scale = do_scale;
    } else {
      hipLaunchKernelGGL((k_x_rowstats<false, true>), g_rows, dim3(256), 0, s, (const uint8_t *)d_X.p, nslabs, spc, (const int *)nullptr, (unsigned long long *)nullptr, raw);
      hipLaunchKernelGGL(k_x_finish_stats, dim3(g_fin), dim3(256), 0, s, raw, rows, d_f, 4.0 * (double)k, out);
      xp.u = d_f; xp.w = out; xp.a = 4.0 * (double)k;
    }
    MXA_HIP(hipGetLastError());
    post_kind = post;
  }
  if (use_ring && (!post || post_kind)) {
    const int rc = crossprod_to_host_ring((const uint8_t *)d_X.p, k, rows, pitch, ans, s, f4, post_kind, &xp, start_prefault, &prefault);
    clk.mark("slabs computed and copied out (ring)");
    d_X.release();
    clk.mark("device buffers released");
    return rc;
  }
  if (use_ring) {   // unfused post-processing needs the whole matrix on the device after all
    if (alloc_result(d_out, abytes)) return 1;
    d_ans = (double *)d_out.p;
    start_prefault();
  }
  if (!out_dev && (!post || post_kind) && whole && ld == rows && !getenv("MXA_XPROD_NO_PIPELINE")) {
    const int rc = crossprod_to_host((const uint8_t *)d_X.p, k, rows, pitch, d_ans, ans, s, f4, post_kind, &xp, &prefault);
    clk.mark("tiles computed, slabs copied out");
    d_out.release(); d_X.release();
    clk.mark("device buffers released");
    return rc;
  }
  if (crossprod_device((const uint8_t *)d_X.p, k, rows, pitch, d_ans, s, c_begin, c_end, upper_only, ld, f4, post_kind, &xp)) return 1;
  if (post && !post_kind && postprocess_device(d_ans, rows, k, post, do_scale, d_f, s)) return 1;
  clk.mark("tile list built, product enqueued");
  if (!out_dev) MXA_HIP(hipMemcpyAsync(ans, d_ans, abytes, hipMemcpyDeviceToHost, s));
  MXA_HIP(hipStreamSynchronize(s));
  clk.mark("product (and download) finished");
  return 0;
}

}  // namespace mxa

extern "C" int snp_multiply_gpu(unsigned char *snp_matrix, int snps, int indiv, double *ans, bool is_plink_format) {
  // positional meaning as in the reference: arg 2 = packed (inner) dimension, arg 3 = output dimension
  mxa::clear_error();
  // MIRACULIX_NUM_GPUS = G > 1 with host operands: output-tile sharding inside this process (packed matrix replicated,
  // independent units, no collective).  Device g stages X itself, computes the column panel [c_g, c_g+1) of the symmetric result -- equal
  // numbers of 256-column tiles, which is equal work: a panel of t tile columns touches t * (rows / 256) tiles -- and downloads it over
  // its own PCIe link into the contiguous slab ans + c_g * indiv of the column-major host matrix.  Every tile is computed by the same
  // kernel as on one device, so the result is bit-identical.  Total work is the full matrix (2x the triangular single-device launch):
  // at config 3 the 80 GB download, not the arithmetic, bounds a host result, and that is what the G links divide.
  const int G = mxa::multi_requested();
  if (G > 1 && snp_matrix && ans && snps > 0 && indiv > 0 && mxa::ptr_location(snp_matrix, nullptr) == 0 && mxa::ptr_location(ans, nullptr) == 0) {
    const long nb = ((long)indiv + mxa::kXT - 1) / mxa::kXT;
    const int parts = (int)std::min<long>(G, nb);
    const int ndev = mxa_device_count();
    return mxa::run_on_devices(parts, [&](int g, int dev) {
      mxa::tl_xprod_shared_device = ndev > 0 && parts > ndev;   // worker threads are pooled: set on every job
      const long c0 = std::min<long>(indiv, nb * g / parts * mxa::kXT), c1 = std::min<long>(indiv, nb * (g + 1) / parts * mxa::kXT);
      if (c1 <= c0) return 0;
      return mxa::crossprod_any(snp_matrix, snps, indiv, ans + (size_t)c0 * indiv, is_plink_format, 0, 0, nullptr, c0, c1, false, indiv, dev);
    });
  }
  return mxa::crossprod_any(snp_matrix, snps, indiv, ans, is_plink_format);
}

extern "C" int mxa_snp_multiply_panel(const unsigned char *snp_matrix, int snps, int indiv, int col_begin, int col_end, int upper_only, double *panel,
                                      long ld, int is_plink_format) {
  mxa::clear_error();
  return mxa::crossprod_any(snp_matrix, snps, indiv, panel, is_plink_format != 0, 0, 0, nullptr, col_begin, col_end, upper_only != 0, ld);
}

extern "C" int mxa_grm(const unsigned char *plink_transposed, int snps, int indiv, double *G, int is_plink_format, int do_scale, const double *allele_freq) {
  mxa::clear_error();
  if (do_scale && !allele_freq) { mxa::set_error(1, "mxa_grm: allele frequencies are required when do_scale is set"); return 1; }
  return mxa::crossprod_any(plink_transposed, snps, indiv, G, is_plink_format != 0, 1, do_scale, allele_freq);
}

extern "C" int mxa_ld(const unsigned char *plink, int snps, int indiv, double *R, int is_plink_format, const double *allele_freq) {
  mxa::clear_error();
  if (!allele_freq) { mxa::set_error(1, "mxa_ld: allele frequencies are required"); return 1; }
  return mxa::crossprod_any(plink, indiv, snps, R, is_plink_format != 0, 2, 0, allele_freq);
}
