// mxa_kernels.hip -- hand-written gfx950 (CDNA4) kernels of the compressed-genotype GEMM path.
//
//   k_recode        PLINK .bed bytes -> padded z-coded 2-bit rows (one-off staging pass, HBM-bound)
//   k_pack_B        B (column-major, ldb) -> MFMA fragment order, zero padded (HBM-bound, once per call)
//   k_colsum_*      deterministic column sums  sum_k B[k,j]  and  sum_k f_k B[k,j]  (centring coefficients)
//   k_gemm          the hot loop: 2-bit genotypes x fp64 on v_mfma_f64_4x4x4_4b_f64 (fp64-MFMA-bound)
//   k_finish        fixed-order split-K reduction + rank-1 centring + store with ldc
//   k_transpose_2bit / k_allele_freq   on-device staging helpers
//
// What they replace in the reference: the CUTLASS SIMT u2 x f64 GEMM (src/cuda/dgemm_compressed_cuda.cu:328-385,
// microkernel src/cuda/dgemm_compressed_cuda.h:232-269), the cublasDgeam transpose (:392-404), the
// Dgemv + n x Daxpy centring epilogue (:421-463) and the per-call memset/2D-memcpy padding of B (:296-317).
//
// gfx950 only; wave64; no CUDA compatibility layer.
#include "mxa_internal.h"
#include "mxa_queue.h"
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

namespace mxa {

// =====================================================================================================
// staging: recode
// =====================================================================================================
// PLINK code c -> allele count z = max(c-1,0): 00->00, 01->00 (missing), 10->01, 11->10, SWAR on 16 fields.
constexpr long kMaxBlocksPerLaunch = 1L << 23;   // x 256 threads = 2^31 threads per launch (the runtime's limit is 2^32 - 1)

__device__ __forceinline__ uint32_t recode16(uint32_t w) {
  const uint32_t H = (w >> 1) & 0x55555555u, L = w & 0x55555555u;
  return ((H & L) << 1) | (H & ~L);
}

// Workgroup = 32 rows x 4 slabs (128 source bytes per row): 8 threads read one full 128-byte run of a source row (16 bytes each,
// dword loads when the address allows), and the 64 threads of a slab write one contiguous 1 KiB run of the tiled layout
// (mxa_internal.h: byte b of row R -> ((R/256)*nslabs + b/32)*8192 + (R%256)*32 + b%32).
__global__ void __launch_bounds__(256) k_recode(const uint8_t *__restrict__ src, size_t src_pitch, long src_row_bytes,
                                                long nrows, long k, uint8_t *__restrict__ dst, long dst_row0, long nslabs, long rows_blocks, long blk0) {
  const long bid = blk0 + blockIdx.x;
  const long rblk = bid % rows_blocks, sgrp = bid / rows_blocks;
  const int rr = threadIdx.x >> 3, part = threadIdx.x & 7;
  const long r = rblk * 32 + rr;
  const long slab = sgrp * 4 + (part >> 1);
  if (r >= nrows || slab >= nslabs) return;
  const long b = slab * kSlabBytes + (part & 1) * 16;        // first source byte of this thread's 16
  uint32_t w[4] = {0u, 0u, 0u, 0u};
  const uint8_t *p = src + (size_t)r * src_pitch + b;
  if (b + 16 <= src_row_bytes && (reinterpret_cast<size_t>(p) & 3) == 0) {
    const uint32_t *q = reinterpret_cast<const uint32_t *>(p);
    w[0] = q[0]; w[1] = q[1]; w[2] = q[2]; w[3] = q[3];
  } else {
    for (int u = 0; u < 16; u++)
      if (b + u < src_row_bytes) w[u >> 2] |= (uint32_t)p[u] << (8 * (u & 3));
  }
#pragma unroll
  for (int i = 0; i < 4; i++) {
    uint32_t x = recode16(w[i]);
    const long valid = k - 4 * (b + 4 * i);                  // genotypes of this dword that exist
    if (valid <= 0) x = 0;
    else if (valid < 16) x &= (1u << (2 * valid)) - 1u;
    w[i] = x;
  }
  const long R = dst_row0 + r;
  const size_t off = ((size_t)(R / kTileRows) * nslabs + (size_t)slab) * kTileBytes + (size_t)(R % kTileRows) * kSlabBytes + (part & 1) * 16;
  *reinterpret_cast<uint4 *>(dst + off) = make_uint4(w[0], w[1], w[2], w[3]);
}

int launch_recode(const uint8_t *d_src, size_t src_pitch, long row0, long nrows, long k, long /*unused*/,
                  PackedMatrix &dst, hipStream_t s) {
  if (nrows <= 0) return 0;
  const long rows_blocks = (nrows + 31) / 32, slab_groups = (dst.nslabs + 3) / 4;
  const long grid = rows_blocks * slab_groups;
  // a launch may not exceed 2^32 threads (grid x block): larger grids are silently truncated by the runtime -> chunks of 2^23 blocks
  for (long b0 = 0; b0 < grid; b0 += kMaxBlocksPerLaunch) {
    const long nb = std::min(kMaxBlocksPerLaunch, grid - b0);
    hipLaunchKernelGGL(k_recode, dim3((unsigned)nb), dim3(256), 0, s, d_src, src_pitch, (k + 3) / 4, nrows, k, dst.d, row0, dst.nslabs, rows_blocks, b0);
  }
  MXA_HIP(hipGetLastError());
  return 0;
}

// =====================================================================================================
// B -> fragment order
// =====================================================================================================
// Bp[chunk][S][h][l]: chunk = column chunk of 4*C columns (one per k_gemm column pass), S = K-step of 16, h = group of 4
// columns inside the chunk (h < C), l = lane 0..63 with j = l&3 (column in group), kk = l>>2 (row in K-step):
// element B[16S+kk][chunk*4C + 4h + j], zero outside k x n.  One v_mfma_f64_4x4x4_4b_f64 takes the 64 doubles of (S,h) as
// its B operand, lane l <- Bp[..][S][h][l] (lane map measured on gfx950: B lane = j + 4*blk + 16*k, we assign genotype row
// 16S + (blk + 4*k) = 16S + (l>>2) to it; tools/mfma_f64_probe2.hip).  A slab of 8 K-steps of one chunk is one contiguous
// run of C*4 KiB, so it streams HBM -> LDS as C*4 lane-linear LDS-DMA units for any C.
// K-steps [S0, S0 + S_cnt) only (the host-operand pipeline packs B in K ranges as they arrive); the whole array: S0 = 0, S_cnt = S_total.
// K ORDER of k_gemm (round 5).  K-step S = 8 slab + t (t = 0..7) of the MFMA stream, K slot kk = 0..15 of the instruction (lane >> 2)  <->  genotype
//   c = 128 slab + 16 (kk & 7) + 8 (kk >> 3) + t.
// A lane's eight K-steps of a slab then take their genotypes from ONE dword of a packed row -- dword kk & 7, fields 8 (kk >> 3) + t -- so the plain form
// (output rows = packed rows) reads one dword per row and slab (A / 2 ds_read2_b32 per slab) where the natural order c = 16 S + kk needed the row's
// whole 32 bytes (2 A reads of 16 bytes; every lane used one field in sixteen).  The transposed-operand form keeps its LDS reads: the permutation sits on
// the global side of its LDS-DMA (k_gemm: tr_voff).  Both forms and k_pack_B share this one map, so they still produce identical sums.
__host__ __device__ __forceinline__ long gemm_k_index(long S, int kk) { return (S >> 3) * 128 + 16 * (kk & 7) + 8 * (kk >> 3) + (S & 7); }

__global__ void __launch_bounds__(256) k_pack_B(const double *__restrict__ B, long ldb, long k, int n,
                                                double *__restrict__ Bp, long total, int C, long S_total, const int *__restrict__ E, int up, long S0, long S_cnt,
                                                const int *__restrict__ run_if_set, int rowscale) {
  if (run_if_set && *run_if_set == 0) return;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int l = (int)(idx & 63);
    const long sh = idx >> 6;
    const int h = (int)(sh % C);
    const long cs = sh / C;
    const long S = S0 + cs % S_cnt;
    const int chunk = (int)(cs / S_cnt);
    const long row = gemm_k_index(S, l >> 2);
    const int col = chunk * 4 * C + 4 * h + (l & 3);
    double v = 0.0;
    if (row < k && col < n) {
      v = B[row + (long)col * ldb];
      // denormal-operand mode: column scaled to just below 2^up (exact); rowscale (k_gemm MODE 3, plain form): row c also by 4^-(c & 15), because
      // the genotype operand of that row is z * 4^(c & 15) (the field is masked where it stands in its dword)
      if (E) v = ldexp(v, up - E[col] - (rowscale ? 2 * (int)(row & 15) : 0));
    }
    Bp[(((size_t)chunk * S_total + S) * C + h) * 64 + l] = v;
  }
}

// ---- per-column binary exponent: E[j] = e + bias with max_k |B[k,j]| = f * 2^e, f in [0.5, 1)  (0 + bias for a zero or non-finite column)
// part[j*64 + c] = max |b| of chunk c (inf if a non-finite entry was seen); part[(n + j)*64 + c] = smallest NON-ZERO |b| of the chunk
// (inf if there is none) when want_min.
__global__ void __launch_bounds__(256) k_colmax_partial(const double *__restrict__ B, long ldb, long k, double *__restrict__ part, int n, int want_min) {
  const int j = blockIdx.y, c = blockIdx.x;
  const long per = (k + 63) / 64;
  const long c0 = c * per, c1 = std::min<long>(k, c0 + per);
  const double inf = __longlong_as_double(0x7ff0000000000000ll);
  double m = 0.0, lo = inf;
  for (long r = c0 + threadIdx.x; r < c1; r += 256) {
    const double a = fabs(B[r + (long)j * ldb]);
    m = (a <= 1.7976931348623157e308) ? fmax(m, a) : inf;   // NaN counts as non-finite too (fmax would drop it)
    if (a > 0.0) lo = fmin(lo, a);
  }
  __shared__ double sh[256], sl[256];
  sh[threadIdx.x] = m; sl[threadIdx.x] = lo;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) { sh[threadIdx.x] = fmax(sh[threadIdx.x], sh[threadIdx.x + w]); sl[threadIdx.x] = fmin(sl[threadIdx.x], sl[threadIdx.x + w]); }
    __syncthreads();
  }
  if (threadIdx.x == 0) { part[(size_t)j * 64 + c] = sh[0]; if (want_min) part[((size_t)n + j) * 64 + c] = sl[0]; }
}
// range guard (flag != nullptr): *flag |= 1 unless every column is finite, its largest exponent e_max >= min_emax, and its smallest
// non-zero entry has exponent e_min >= e_max - max_span (then every entry is an exact multiple of the last digit of the int8 slicing)
__global__ void k_colexp_final(const double *__restrict__ part, int n, int bias, int *__restrict__ E, int *__restrict__ flag, int max_span, int min_emax) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  double m = 0.0;
  for (int c = 0; c < 64; c++) m = fmax(m, part[(size_t)j * 64 + c]);
  int e = 0;
  if (m > 0.0 && isfinite(m)) (void)frexp(m, &e);
  E[j] = e + bias;
  if (flag) {
    bool ok = isfinite(m);
    if (ok && m > 0.0) {
      double lo = m;
      for (int c = 0; c < 64; c++) lo = fmin(lo, part[((size_t)n + j) * 64 + c]);
      int el = 0;
      (void)frexp(lo, &el);
      ok = e >= min_emax && el >= e - max_span;
    }
    if (!ok) atomicOr(flag, 1);
  }
}
int launch_colexp(const double *dB, long ldb, long k, int n, double *d_part, int *d_E, int bias, hipStream_t s, int *d_flag, int max_span, int min_emax, bool reset_flag) {
  if (d_flag && reset_flag) MXA_HIP(hipMemsetAsync(d_flag, 0, sizeof(int), s));
  hipLaunchKernelGGL(k_colmax_partial, dim3(64, n), dim3(256), 0, s, dB, ldb, k, d_part, n, d_flag ? 1 : 0);
  hipLaunchKernelGGL(k_colexp_final, dim3((n + 63) / 64), dim3(64), 0, s, d_part, n, bias, d_E, d_flag, max_span, min_emax);
  MXA_HIP(hipGetLastError());
  return 0;
}

// exponents only, from column maxima that are already there (launch_colspan ran over the same columns)
int launch_colexp_from_part(const double *d_part, int n, int *d_E, int bias, hipStream_t s) {
  hipLaunchKernelGGL(k_colexp_final, dim3((n + 63) / 64), dim3(64), 0, s, d_part, n, bias, d_E, (int *)nullptr, 0, 0);
  MXA_HIP(hipGetLastError());
  return 0;
}

// engine 4 (i8-exact): what the host needs to choose the digit count: out[0] = largest exponent span e_max - e_min over the columns (non-zero
// entries only), out[1] = smallest e_max over the non-zero columns (0 if there is none), out[2] = 1 if any entry is inf / NaN.  One workgroup.
__global__ void __launch_bounds__(256) k_colspan_final(const double *__restrict__ part, int n, int *__restrict__ out) {
  __shared__ int s_span[256], s_emax[256], s_bad[256];
  int span = 0, emin_max = 1 << 20, bad = 0;
  for (int j = threadIdx.x; j < n; j += 256) {
    double m = 0.0;
    for (int c = 0; c < 64; c++) m = fmax(m, part[(size_t)j * 64 + c]);
    if (!isfinite(m)) { bad = 1; continue; }
    if (m > 0.0) {
      double lo = m;
      for (int c = 0; c < 64; c++) lo = fmin(lo, part[((size_t)n + j) * 64 + c]);
      int e = 0, el = 0;
      (void)frexp(m, &e); (void)frexp(lo, &el);
      span = max(span, e - el); emin_max = min(emin_max, e);
    }
  }
  s_span[threadIdx.x] = span; s_emax[threadIdx.x] = emin_max; s_bad[threadIdx.x] = bad;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) {
      s_span[threadIdx.x] = max(s_span[threadIdx.x], s_span[threadIdx.x + w]); s_emax[threadIdx.x] = min(s_emax[threadIdx.x], s_emax[threadIdx.x + w]);
      s_bad[threadIdx.x] |= s_bad[threadIdx.x + w];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) { out[0] = s_span[0]; out[1] = s_emax[0] == (1 << 20) ? 0 : s_emax[0]; out[2] = s_bad[0]; }
}
int launch_colspan(const double *dB, long ldb, long k, int n, double *d_part, int *d_out3, hipStream_t s) {
  hipLaunchKernelGGL(k_colmax_partial, dim3(64, n), dim3(256), 0, s, dB, ldb, k, d_part, n, 1);
  hipLaunchKernelGGL(k_colspan_final, dim3(1), dim3(256), 0, s, d_part, n, d_out3);
  MXA_HIP(hipGetLastError());
  return 0;
}

int launch_pack_B(const double *dB, long ldb, long k, int n, double *dBp, long k_pad, int n_pad, int c, hipStream_t s, const int *d_E, long S0, long S_cnt,
                  const int *run_if_set, bool rowscale) {
  const long S_total = k_pad / 16;
  if (S_cnt < 0) S_cnt = S_total - S0;
  const long total = S_cnt * (long)(n_pad / 4) * 64;
  if (total <= 0) return 0;
  const int grid = (int)std::min<long>((total + 255) / 256, 256L * 64);
  hipLaunchKernelGGL(k_pack_B, dim3(grid), dim3(256), 0, s, dB, ldb, k, n, dBp, total, c, S_total, d_E, kDenUp, S0, S_cnt, run_if_set, rowscale ? 1 : 0);
  MXA_HIP(hipGetLastError());
  return 0;
}

// =====================================================================================================
// deterministic column sums
// =====================================================================================================
constexpr int kColChunks = 64;

__global__ void __launch_bounds__(256) k_colsum_partial(const double *__restrict__ B, long ldb, long k,
                                                        const double *__restrict__ f, double *__restrict__ part) {
  // grid (kColChunks, n): block sums rows [c0, c1) of column j; fixed thread->row assignment and fixed tree
  const int j = blockIdx.y, c = blockIdx.x;
  const long per = (k + kColChunks - 1) / kColChunks;
  const long c0 = c * per, c1 = std::min<long>(k, c0 + per);
  double s1 = 0.0, s2 = 0.0;
  for (long r = c0 + threadIdx.x; r < c1; r += 256) {
    const double b = B[r + (long)j * ldb];
    s1 += b;
    if (f) s2 = fma(f[r], b, s2);
  }
  __shared__ double sh1[256], sh2[256];
  sh1[threadIdx.x] = s1; sh2[threadIdx.x] = s2;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) { sh1[threadIdx.x] += sh1[threadIdx.x + w]; sh2[threadIdx.x] += sh2[threadIdx.x + w]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    part[((size_t)j * kColChunks + c) * 2 + 0] = sh1[0];
    part[((size_t)j * kColChunks + c) * 2 + 1] = sh2[0];
  }
}

__global__ void k_colsum_final(const double *__restrict__ part, int n, double *__restrict__ sumB, double *__restrict__ sumfB) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  double s1 = 0.0, s2 = 0.0;
  for (int c = 0; c < kColChunks; c++) { s1 += part[((size_t)j * kColChunks + c) * 2]; s2 += part[((size_t)j * kColChunks + c) * 2 + 1]; }
  sumB[j] = s1; sumfB[j] = s2;
}

int launch_colsums(const double *dB, long ldb, long k, int n, const double *d_f, double *d_part, double *d_sumB,
                   double *d_sumfB, hipStream_t s) {
  hipLaunchKernelGGL(k_colsum_partial, dim3(kColChunks, n), dim3(256), 0, s, dB, ldb, k, d_f, d_part);
  hipLaunchKernelGGL(k_colsum_final, dim3((n + 63) / 64), dim3(64), 0, s, d_part, n, d_sumB, d_sumfB);
  MXA_HIP(hipGetLastError());
  return 0;
}

// =====================================================================================================
// the hot loop
// =====================================================================================================
// Work decomposition.  C(m x n) = G(m x k, 2-bit) * B(k x n).  A workgroup (4 waves) owns 4*A*4 rows x 4*C columns
// for one K range; wave w owns rows [w*4A, (w+1)*4A).  Per K-step of 16 genotypes a wave issues A*C
// v_mfma_f64_4x4x4_4b_f64:  acc[g][h] (4 rows x 4 cols, K split over the instruction's 4 blocks)
//   += Ablk(4 rows x 16 k) * Bblk(16 k x 4 cols).
// The 16x16x4 f64 MFMA issues at ~61 % of the fp64 peak on gfx950 (47.8 of 78.6 TFLOP/s measured), the 4-block
// 4x4x4 form at 96 % (75.8 TFLOP/s): tools/mfma_f64_probe*.hip, profiles/r01_mfma_f64_probe.txt.
// Lane maps (measured): A lane = i + 4*blk + 16*kq (row i, K index blk + 4*kq =: lane>>2), B lane = j + 4*blk + 16*kq,
// D lane = j + 4*blk + 16*i.  The 4 blocks hold partial sums over disjoint K subsets; they are added once, after
// the K loop, by two cross-lane adds.
//
// Data path.  Everything global comes in by LDS-DMA (global_load_lds_dwordx4): per slab of 128 genotypes the B
// fragments (8 K-steps x C groups x 512 B, lane-linear, so they stream straight in) and the packed genotype rows
// (32 B per row).  Two LDS buffers; one vmcnt(0)+barrier per slab.  In the compute phase a wave only issues
// ds_read_b64, 4 integer VALU per genotype fragment (VALU time is NOT hidden beside the fp64 MFMA: the DP pipe is
// shared) and MFMAs.
using gptr_t = const __attribute__((address_space(1))) void *;
using lptr_t = __attribute__((address_space(3))) void *;

// LDS-DMA with a wave-uniform 64-bit base in SGPRs, a per-lane 32-bit byte offset and a wave-uniform LDS byte address
// (M0).  Written as asm so that the per-slab address arithmetic stays on the scalar unit; hipcc does not count this
// load: the kernel waits with its own s_waitcnt vmcnt(0) before the barrier that precedes the first ds_read of the data.
__device__ __forceinline__ void dma16_s(const void *sbase, uint32_t voff, uint32_t lds_addr) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %0" ::"s"(sbase), "v"(voff), "s"(lds_addr) : "memory", "m0");
}
// non-temporal variant for the packed genotype stream (read once per pass, far larger than every cache): see mxa_gemm_i8.hip, idma16_stream.
// Measured on the config-5 shard (profiles/r03_nt_stream_ab.txt): k_gemm_i8 at n = 1 0.96-0.97 ms with the hint against 1.01-1.02 without; the lookup kernel
// k_lut 1.66 against 1.63 (LDS-bound: no gain, MXA_NT_LUT stays 0).  k_gemm is MFMA-bound and its time does not move (C2: 43.89-43.99 ms with, 43.87-44.02
// without), but its HBM traffic does: the packed stream no longer pushes the B-fragment slabs, which every row block re-reads, out of the L2s --
// 15.0 GB per launch by the counters instead of 18.9 (12.8 algorithmic; FETCH_SIZE calibrated for the hinted stream too: factor 2.000,
// profiles/r03_pmc_calibration_nt.json).  MXA_NT_GEMM = 1.
#ifndef MXA_NT_LUT
#define MXA_NT_LUT 0
#endif
#ifndef MXA_NT_GEMM
#define MXA_NT_GEMM 1
#endif
template <bool NT>
__device__ __forceinline__ void dma16_p(const void *sbase, uint32_t voff, uint32_t lds_addr) {
  if (NT) asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %0 nt" ::"s"(sbase), "v"(voff), "s"(lds_addr) : "memory", "m0");
  else dma16_s(sbase, voff, lds_addr);
}

template <int A, int C>
struct GemmCfg {
  static constexpr int kRowsWave = 4 * A;
  static constexpr int kRowsWG = kWaves * kRowsWave;
  static constexpr int kBBytes = kSlabSteps * C * 512;      // B fragments of one slab
  static constexpr int kABytes = kRowsWG * kSlabBytes;      // packed rows of one slab
  static constexpr int kBufBytes = kBBytes + kABytes;
  static constexpr int kLds = 2 * kBufBytes;
  static constexpr int kBUnits = kBBytes / 1024;            // 1 KiB wave-instruction units (= 4C, C per wave)
  static constexpr int kAUnits = kABytes / 1024;
  static_assert(kABytes % 1024 == 0, "A tile must be a whole number of DMA units");
};

// Conversion of a 2-bit allele count z (lane's field of the packed word) to the fp64 MFMA operand, exact in all modes:
//   MODE 2 (shipped): v_bfe_u32 only.  The register pair (low word z, high word 0) IS the double z * 2^-1074, a denormal, which
//                     v_mfma_f64_4x4x4_4b_f64 takes exactly and at full rate (tools/mfma_f64_denorm_probe.hip: bit-identical
//                     to normal operands over 4096 accumulation steps, 76.5 vs 75.1 TFLOP/s).  The high words are zeroed once
//                     before the K loop and stay in place; k_pack_B scales every column of B to just below 2^900 (per-column
//                     exponent from launch_colexp), k_finish scales the sums back by 2^(174 + e_j): powers of two, so every
//                     product and sum is the same as with normal operands unless an intermediate would under- or overflow.
//   MODE 0:           v_bfe_u32 + v_cvt_f64_u32 -- 2 VALU per fragment, all A fragments of a K-step converted first, then
//                     the A*C MFMAs back to back (pinned with sched_barrier).
//   MODE 3:           as MODE 2 with v_and_b32 (VOP2) instead of v_bfe_u32 (VOP3): the field is masked where it stands, the power-of-two scale
//                     of its position is divided out exactly (plain form: k_pack_B scales the rows of B; transposed form: the epilogue).
//   (MODE 1, the high word built by four integer VALU instructions, was removed in round 5: 63.6 TFLOP/s against 69.7 for MODE 0.)
// Why it matters: on gfx950 every VALU instruction in an fp64-MFMA stream costs 6-13 cycles of MFMA time, wherever it
// is placed and whether or not an MFMA depends on it (tools/mfma_f64_probe5.hip, profiles/r01_mfma_f64_probe.txt): the
// VALU instruction count per MFMA is the lever, not latency hiding.  Measured on MI355X, 200k x 50k x 32:
// MODE 1 63.6 TFLOP/s, MODE 0 69.7 TFLOP/s; with the LDS-DMA addresses computed on the VALU instead of the scalar
// unit MODE 0 drops to 65.5.  MODE 2 vs MODE 0 at 1M x 50k: n = 32 (C = 8) 17.1k vs 17.3k cycles per slab (+1 %),
// n = 20 (C = 5) 72.0 vs 68.7 TFLOP/s, n = 10 (C = 3) 56.0 vs 52.0.
// PERSISTENT workgroups.  The launch has one workgroup per resident slot (2 per CU); each pulls work pieces "unit" = (row block, column chunk,
// K split) from queues in device memory until they are empty.  Why (MXA_DIAG stamps, C2): with one workgroup per piece the hardware needs
// 8-14 us to replace a finished workgroup (0.36-0.42 ms of idle slot time per 44.6 ms launch) and every piece begins with an exposed first
// slab load (2.8 us); here the next piece's first slab is in flight during the epilogue of the current one.
// Queues: pieces are numbered like the block index of the former one-piece-per-workgroup launch -- row block fastest inside a (column chunk,
// K split) group, whose workgroups stream the same B slabs.  With xcd_order the first 8 * floor(groups / 8) groups are dealt whole to the 8
// XCDs (each has its own L2: group g is worked on by XCD g % 8 only, its B slabs cross the fabric once instead of once per XCD): queue x
// holds the pieces of XCD x in order, queue 8 the remaining groups in plain order for everybody; a workgroup whose queues are empty steals
// from the other XCDs' queues.  The XCD of a workgroup is read from the hardware (HW_REG_XCC_ID), not assumed.  ctr: 9 counters, zeroed by
// the launcher on the same stream.  Every fetch is one returning atomic add by one lane; every wave leaves when the fetch returns "none".
// TR (round 4, transposed operand): the OUTPUT rows are the packed matrix's COLUMNS and K runs over its ROWS -- the 'N' product C = Z B computed from the
// SNP-major copy (rows = SNPs, 32-byte slab pieces of 128 individuals), so that one stored orientation can serve both products.  A workgroup then owns
// kRowsWG individuals = kRowsWG / 128 slabs; a K slab is 128 SNP rows = half a 256-row tile (4 KiB contiguous per slab of individuals); lane (i, kk) reads
// the wave's 8 / 16 bytes of packed row 16 S + kk (ds_read_b64 / b128; the four lanes of a K index read the same word) and extracts the fields of its
// individuals 4 g + i: still ONE v_bfe_u32 per fragment, with a per-lane shift 8 (g & 3) + 2 i.  Same K order, same partial sums, same P layout as the
// untransposed launch on the individual-major copy: bit-identical results.  MODE 3 here: the field is masked where it stands (v_and_b32) and the scale
// 4^(field) belongs to the OUTPUT row, undone exactly in the epilogue (no row scaling of B).
template <int A, int C, int MODE, bool DIAG = false, bool TR = false>
__global__ void __launch_bounds__(256, 2)
k_gemm(const uint8_t *__restrict__ G, size_t pitch, const double *__restrict__ Bp, int H, double *__restrict__ P,
       long m_pad, int n_pad, int rowblocks, int nchunks, int slabs_total, KSplit ks, int xcd_order,
       unsigned long long *__restrict__ diag, int split0, const int *__restrict__ run_if_set, int nunits, int *__restrict__ ctr, int g8, int p_split0) {
  using Cfg = GemmCfg<A, C>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __shared__ int s_next;
  // MODE 0 as the fallback of the denormal-operand mode: runs only when launch_colexp found a column outside its range (kDenMaxSpan)
  if (MODE == 0) { if (run_if_set && *run_if_set == 0) return; }

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // g8 (a multiple of 8, from the launcher): groups dealt to the XCDs whole -- LONG pieces only, the same number for every XCD, so that
  // the XCD queues hold equal work; the remaining long groups and the short ones (the tapered tail) form the common queue (mxa_queue.h)
  PieceQueue q;
  q.init(ctr, rowblocks, g8, nunits);
  auto fetch = [&]() -> int { return q.next(); };

  // ---- DMA issue for one slab into buffer `buf`.  Every source address is (wave-uniform 64-bit base) + (per-lane 32-bit
  // offset that never changes), so the per-slab address arithmetic is scalar: VALU instructions are expensive beside the
  // fp64 MFMA stream (see MODE comment above).
  const uint32_t b_lane = lane * 16;
  const uint32_t a_lane = lane * 16;
  // TR: lane l of a 1 KiB unit fills LDS row position P = 32 (q & 3) + (l >> 1), half l & 1: K-step t = 2 (q & 3) + (l >> 5), K slot kk = (l >> 1) & 15;
  // source = packed row 16 (kk & 7) + 8 (kk >> 3) + t of the half tile (32 bytes per row); the 2 (q & 3) rows are the scalar part of the address
  const uint32_t tr_voff = (uint32_t)((16 * ((lane >> 1) & 7) + 8 * ((lane >> 4) & 1) + (lane >> 5)) * kSlabBytes + (lane & 1) * 16);
  const size_t nslabs_all = pitch / kSlabBytes;
  const uint32_t lds0 = (uint32_t)(size_t)(lptr_t)smem;
  struct Unit { int rb, nc, sp, slab0, slab1; const char *Bp_u, *G_u; };
  auto decode = [&](int bid) -> Unit {
    Unit u;
    int grp;
    q.locate(bid, u.rb, grp);
    u.nc = grp % nchunks;
    u.sp = grp / nchunks + split0;                   // split0: first K split of this launch (host-operand pipeline: K ranges as B arrives)
    u.slab0 = ksplit_begin(ks, u.sp);
    u.slab1 = u.slab0 + ksplit_len(ks, u.sp);
    const long row0 = (long)u.rb * Cfg::kRowsWG;
    // chunk-major B fragments: the column chunk nc is one contiguous array [S_total][C][64]
    u.Bp_u = reinterpret_cast<const char *>(Bp) + (size_t)u.nc * ((size_t)slabs_total * kSlabSteps * C * 512);
    // tiled layout: the rows [row0, row0 + kRowsWG) of slab s are one contiguous run inside tile (row0/256, s)
    if (!TR) u.G_u = reinterpret_cast<const char *>(G) + (size_t)(row0 / kTileRows) * nslabs_all * kTileBytes + (size_t)(row0 % kTileRows) * kSlabBytes;
    else u.G_u = reinterpret_cast<const char *>(G);   // TR: the workgroup's slabs of individuals are addressed per DMA unit (issue)
    return u;
  };
  auto issue = [&](const Unit &u, int slab, int buf) {
    const uint32_t base = lds0 + buf * Cfg::kBufBytes;
    // B fragments of the slab: one contiguous run of C*4 KiB -> C units per wave
    const char *bslab = u.Bp_u + (size_t)slab * ((size_t)kSlabSteps * C * 512);
#pragma unroll
    for (int i = 0; i < Cfg::kBUnits / kWaves; i++) {
      const int q = wave + i * kWaves;
      dma16_s(bslab + q * 1024, b_lane, base + q * 1024);
    }
    // packed genotype rows: unit = 32 rows x 32 B = 1 KiB, contiguous in the tiled layout
    if (!TR) {
      const char *aslab = u.G_u + (size_t)slab * kTileBytes;
#pragma unroll
      for (int i = 0; i < (Cfg::kAUnits + kWaves - 1) / kWaves; i++) {
        const int q = wave + i * kWaves;
        if (Cfg::kAUnits % kWaves == 0 || q < Cfg::kAUnits) dma16_p<MXA_NT_GEMM != 0>(aslab + q * 1024, a_lane, base + Cfg::kBBytes + q * 1024);
      }
    } else {
      // K slab `slab` = packed rows [128 slab, 128 slab + 128): the half (slab & 1) of tile row slab >> 1; the workgroup's j-th slab of individuals is tile
      // column u.rb * nsw + j (clamped to the last one: individuals beyond the matrix give rows of P that nobody reads)
      constexpr int nsw = Cfg::kRowsWG / kSlabK;
      const char *arow = u.G_u + (size_t)(slab >> 1) * nslabs_all * kTileBytes + (size_t)(slab & 1) * (kTileBytes / 2);
#pragma unroll
      for (int i = 0; i < (Cfg::kAUnits + kWaves - 1) / kWaves; i++) {
        const int q = wave + i * kWaves;
        if (Cfg::kAUnits % kWaves == 0 || q < Cfg::kAUnits) {
          size_t sl = (size_t)u.rb * nsw + (q >> 2);
          if (sl >= nslabs_all) sl = nslabs_all - 1;
          // LDS row position P = 32 (q & 3) + lane / 2 = 16 t + kk receives packed row gemm_k_index(t, kk) of the K slab (tr_voff: the lane's part)
          dma16_p<MXA_NT_GEMM != 0>(arow + sl * kTileBytes + (q & 3) * 64, tr_voff, base + Cfg::kBBytes + q * 1024);
        }
      }
    }
  };

  typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
  u32x2 ap[A];                                                    // MODE 2 / 3: A operands as (low, high) word pairs, high words stay 0
#pragma unroll
  for (int g = 0; g < A; g++) {
    uint32_t zero;   // opaque to the optimiser: a known constant would be re-materialised next to every low word (one v_mov per fragment)
    asm volatile("v_mov_b32 %0, 0" : "=v"(zero));
    ap[g].x = 0; ap[g].y = zero;
  }
  // plain form: the lane's dword of a packed row, and its field in K-step kst: 8 (lane >> 5) + kst (gemm_k_index)
  uint32_t psh[kSlabSteps], pmask[kSlabSteps];
#pragma unroll
  for (int t = 0; t < kSlabSteps; t++) {
    psh[t] = 16 * (lane >> 5) + 2 * t;
    pmask[t] = 3u << psh[t];
    asm volatile("" : "+v"(pmask[t]));                            // opaque: kept in a register, not rebuilt by a shift beside the MFMA stream
  }
  const int a_off = (wave * Cfg::kRowsWave + (lane & 3)) * kSlabBytes + ((lane >> 2) & 7) * 4;  // + g*4 rows -> + g*4*32 bytes
  const int b_off = lane * 8;
  // TR: packed row (lane >> 2) of the K-step, the wave's A bytes of it; field shifts of the lane's individuals 4 g + i, g & 3 = 0..3
  const int tr_off = ((wave * A) / kSlabBytes) * (kSlabK * kSlabBytes) + (lane >> 2) * kSlabBytes + (wave * A) % kSlabBytes;
  const int tr_sh[4] = {2 * (lane & 3), 8 + 2 * (lane & 3), 16 + 2 * (lane & 3), 24 + 2 * (lane & 3)};
  const uint32_t tr_mask[4] = {3u << tr_sh[0], 3u << tr_sh[1], 3u << tr_sh[2], 3u << tr_sh[3]};
  const int j = lane & 3, blk = (lane >> 2) & 3, i = lane >> 4;

  if (threadIdx.x == 0) s_next = fetch();
  __syncthreads();
  int bid = s_next;
  if (bid < 0) return;
  Unit u = decode(bid);
  __syncthreads();                                                 // everybody has read s_next before thread 0 writes the next one
  if (u.slab0 < u.slab1) issue(u, u.slab0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  for (;;) {
    unsigned long long rt_begin = 0, t0 = 0, r0 = 0;
    if (DIAG) { rt_begin = __builtin_amdgcn_s_memrealtime(); t0 = __builtin_amdgcn_s_memtime(); r0 = rt_begin; }
    // the piece after this one: fetched now, read by everybody after the K loop (its barriers lie in between)
    if (threadIdx.x == 0) s_next = fetch();
    if (u.slab0 >= u.slab1) __syncthreads();

    double acc[A][C];
#pragma unroll
    for (int g = 0; g < A; g++)
#pragma unroll
      for (int h = 0; h < C; h++) acc[g][h] = 0.0;

    for (int slab = u.slab0; slab < u.slab1; slab++) {
      const int buf = (slab - u.slab0) & 1;
      if (slab + 1 < u.slab1) issue(u, slab + 1, buf ^ 1);
      const char *bbase = smem + buf * Cfg::kBufBytes + b_off;
      const char *abase = smem + buf * Cfg::kBufBytes + Cfg::kBBytes + a_off;
      if (TR) {
        // LDS image [slab of individuals j][128 packed rows][32 B]; the wave's individuals are bytes [tr_off, tr_off + A) of every row
        const char *tbase = smem + buf * Cfg::kBufBytes + Cfg::kBBytes + tr_off;
        constexpr int ND = A / 4;                                    // dwords of a packed row the wave's 4 A individuals occupy
        uint32_t dn[ND];
        auto load_row = [&](int kst, uint32_t (&d)[ND]) {
          if constexpr (ND == 4) { const uint4 t = *reinterpret_cast<const uint4 *>(tbase + kst * 16 * kSlabBytes); d[0] = t.x; d[1] = t.y; d[2] = t.z; d[3] = t.w; }
          else { const uint2 t = *reinterpret_cast<const uint2 *>(tbase + kst * 16 * kSlabBytes); d[0] = t.x; d[1] = t.y; }
        };
        load_row(0, dn);
#pragma unroll
        for (int kst = 0; kst < kSlabSteps; kst++) {
          uint32_t dw[ND];
#pragma unroll
          for (int t = 0; t < ND; t++) dw[t] = dn[t];
          if (kst + 1 < kSlabSteps) load_row(kst + 1, dn);
          double bf[C];
#pragma unroll
          for (int h = 0; h < C; h++) bf[h] = *reinterpret_cast<const double *>(bbase + (kst * C + h) * 512);
          double af[A];
#pragma unroll
          for (int g = 0; g < A; g++) {
            if (MODE == 3) {   // field masked where it stands: the operand is z * 4^(4 (g & 3) + i) * 2^-1074; the epilogue scales row (g, i) back (exact)
              asm volatile("v_and_b32 %0, %1, %2" : "=v"(ap[g].x) : "v"(tr_mask[g & 3]), "v"(dw[g >> 2]));
              af[g] = __builtin_bit_cast(double, ap[g]);
            } else {
              const uint32_t z = __builtin_amdgcn_ubfe(dw[g >> 2], tr_sh[g & 3], 2);
              if (MODE == 0) af[g] = (double)z;
              else { ap[g].x = z; af[g] = __builtin_bit_cast(double, ap[g]); }
            }
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int g = 0; g < A; g++)
#pragma unroll
            for (int h = 0; h < C; h++) acc[g][h] = __builtin_amdgcn_mfma_f64_4x4x4f64(af[g], bf[h], acc[g][h], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
      } else {
        // plain form, permuted K order (gemm_k_index): ONE dword per packed row and slab -- dword (lane >> 2) & 7 of row 4 g + i -- holds the lane's
        // genotypes of all eight K-steps (fields 8 (lane >> 5) + kst)
        uint32_t aw[A];
#pragma unroll
        for (int g = 0; g < A; g++) aw[g] = *reinterpret_cast<const uint32_t *>(abase + g * 4 * kSlabBytes);
#pragma unroll
        for (int kst = 0; kst < kSlabSteps; kst++) {
          double bf[C];
#pragma unroll
          for (int h = 0; h < C; h++) bf[h] = *reinterpret_cast<const double *>(bbase + (kst * C + h) * 512);
          double af[A];
#pragma unroll
          for (int g = 0; g < A; g++) {
            if (MODE == 3) {
              // denormal operand, field masked where it stands (v_and_b32, VOP2): the operand is z * 4^field * 2^-1074 and k_pack_B has scaled
              // row c of B by 4^-(c & 15) (exact).  Only the low word is rewritten; the high words were zeroed once and stay in place.
              asm volatile("v_and_b32 %0, %1, %2" : "=v"(ap[g].x) : "v"(pmask[kst]), "v"(aw[g]));
              af[g] = __builtin_bit_cast(double, ap[g]);
            } else {
              const uint32_t z = __builtin_amdgcn_ubfe(aw[g], psh[kst], 2);
              if (MODE == 0) af[g] = (double)z;
              else { ap[g].x = z; af[g] = __builtin_bit_cast(double, ap[g]); }
            }
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int g = 0; g < A; g++)
#pragma unroll
            for (int h = 0; h < C; h++) acc[g][h] = __builtin_amdgcn_mfma_f64_4x4x4f64(af[g], bf[h], acc[g][h], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }

    unsigned long long t1 = 0, r1 = 0;
    if (DIAG) { t1 = __builtin_amdgcn_s_memtime(); r1 = __builtin_amdgcn_s_memrealtime(); }
    // ---- the next piece's first slab goes into buffer 0 while this piece's results leave through buffer 1 (both buffers are idle: the
    // K loop ended with a barrier)
    const int next = s_next;
    Unit un = u;
    if (next >= 0) { un = decode(next); if (un.slab0 < un.slab1) issue(un, un.slab0, 0); }
    // ---- epilogue: add the 4 K-blocks (lane bits 2..3); lane (j, blk, i) stores column group hq+blk.  The exchange goes through the
    // (now idle) second LDS buffer, not through cross-lane VALU shuffles: while this wave is in its epilogue the other workgroup of the CU
    // is in its MFMA stream, where every VALU instruction of this wave waits for the shared pipe and costs the other one MFMA time
    // (shuffle version: ~330 VALU per lane, 15 us per workgroup; here 3 v_add_f64 per output).
    constexpr int kPitch = 68;                                   // doubles per accumulator image: 64 lanes + pad (2-way conflicts at most)
    constexpr int kPerWave = Cfg::kBufBytes / kWaves / 8;        // doubles of scratch per wave (a quarter of the second buffer)
    constexpr int GP = (kPerWave / (C * kPitch)) < A ? (kPerWave / (C * kPitch)) : A;   // row groups per pass
    static_assert(GP >= 1, "epilogue scratch");
    double *scr = reinterpret_cast<double *>(smem + Cfg::kBufBytes) + wave * kPerWave;
    // P[split][row block][n_pad][rows of the block]: the workgroup's results form one contiguous chunk
    double *Pbase = P + ((size_t)(u.sp - p_split0) * rowblocks + u.rb) * ((size_t)n_pad * Cfg::kRowsWG);   // p_split0: the K split whose partial sums start the buffer (grouped K splits)
#pragma unroll
    for (int g0 = 0; g0 < A; g0 += GP) {
#pragma unroll
      for (int gg = 0; gg < GP; gg++)
#pragma unroll
        for (int h = 0; h < C; h++)
          if (g0 + gg < A) scr[(gg * C + h) * kPitch + lane] = acc[g0 + gg][h];
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");         // the wave's own LDS writes are done before any of its lanes reads them
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int gg = 0; gg < GP; gg++) {
        if (g0 + gg < A) {
          const int row = wave * Cfg::kRowsWave + (g0 + gg) * 4 + i;   // row inside the block
#pragma unroll
          for (int hq = 0; hq < C; hq += 4) {
            if (hq + blk < C) {
              const double *q = scr + (gg * C + hq + blk) * kPitch + j + 16 * i;
              double out = (q[0] + q[4]) + (q[8] + q[12]);
              if (TR && MODE == 3) out = ldexp(out, -(8 * ((g0 + gg) & 3) + 2 * i));   // undo the in-place field scale of output row (g, i): a power of two, exact
              const int col = 4 * (u.nc * C + hq + blk) + j;
              Pbase[(size_t)col * Cfg::kRowsWG + row] = out;   // (a non-temporal store here more than doubles WRITE_SIZE: the 32-byte runs of a wave are no longer merged in the L2)
            }
          }
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_wave_barrier();
    }
    if (DIAG && threadIdx.x == 0 && diag) {   // per piece: cycles / ticks in the K loop, begin, loop start, end (DIAG build only)
      diag[2 * (size_t)bid] = t1 - t0; diag[2 * (size_t)bid + 1] = r1 - r0;
      diag[2 * (size_t)nunits + 2 * (size_t)bid] = rt_begin; diag[2 * (size_t)nunits + 2 * (size_t)bid + 1] = r0;
      diag[4 * (size_t)nunits + bid] = __builtin_amdgcn_s_memrealtime();
    }
    if (next < 0) break;
    u = un; bid = next;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");               // the next piece's first slab has landed (and this piece's stores are out)
    __syncthreads();
  }
}

int gemm_default_mode(int) {
  // MODE 3 (v_and_b32, field masked in place, the scale divided out exactly) for every tile; MODE 0 (plain operands) is the range fallback.  The v_bfe_u32
  // variant (MODE 2) was removed in round 5: slower on every tile in both forms (profiles/r05_gemm_plain_permuted_k_ab.txt: C2 43.2 against 42.4 ms).
  return 3;
}

static long device_cus() {
  static const long r = [] {
    int dev = 0; hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) { (void)hipGetLastError(); return 256L; }
    return (long)prop.multiProcessorCount;
  }();
  return r > 0 ? r : 256;
}

GemmPlan plan_gemm(long m, long k_pad, int n, const GemmPlan *ksplits_like) { return plan_gemm_host(m, k_pad, n, device_cus(), ksplits_like); }

template <int A, int C, int MODE, bool TR = false>
static int launch_gemm_t(const PackedMatrix &G, const double *dBp, double *dP, const GemmPlan &p, hipStream_t s, int split_begin, int split_end, const int *run_if_set,
                         int *d_ctr, int p_split0) {
  using Cfg = GemmCfg<A, C>;
  static unsigned long long attr_mask = 0;   // function attributes are per device
  if (ensure_dyn_lds(reinterpret_cast<const void *>(&k_gemm<A, C, MODE, false, TR>), Cfg::kLds, &attr_mask)) return 1;
  const long nunits = (long)p.rowblocks * p.nchunks * (split_end - split_begin);
  if (nunits > 0x3fffffffL) { set_error(3, "launch too large"); return 1; }
  if (nunits <= 0) return 0;
  if (!d_ctr) { set_error(4, "internal: k_gemm needs its queue counters"); return 1; }
  // persistent workgroups: one per resident slot (the occupancy of this instantiation x the CUs), fewer if there are fewer pieces
  static int per_cu[64] = {};
  int dev = 0;
  MXA_HIP(hipGetDevice(&dev));
  if (!per_cu[dev & 63]) {
    int nb = 0;
    MXA_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void *>(&k_gemm<A, C, MODE, false, TR>), 256, Cfg::kLds));
    hipDeviceProp_t prop;
    MXA_HIP(hipGetDeviceProperties(&prop, dev));
    per_cu[dev & 63] = std::max(1, std::min(nb, gemm_wg_per_cu(C))) * prop.multiProcessorCount;
  }
  const long grid = std::min<long>(nunits, per_cu[dev & 63]);
  constexpr int xcd_order = 1;   // per-XCD queues (one common queue in plain order was the A/B baseline of round 2)
  const KSplit ks{p.s1, p.l1, p.r1, p.l2};
  const int long_groups = (std::min(split_end, p.s1) - std::min(split_begin, p.s1)) * p.nchunks;
  const int g8 = xcd_order ? (long_groups & ~7) : 0;
  MXA_HIP(hipMemsetAsync(d_ctr, 0, 9 * sizeof(int), s));
  static const bool diag_on = getenv("MXA_DIAG") != nullptr;
  if (!TR && diag_on && ((A == 8 && C == 8) || (A == 16 && C == 1)) && split_begin == 0 && split_end == p.splits && !run_if_set) {   // diagnostic instantiation: in-kernel clock + cycles per slab
    static unsigned long long attr2 = 0;
    if (ensure_dyn_lds(reinterpret_cast<const void *>(&k_gemm<A, C, MODE, true>), Cfg::kLds, &attr2)) return 1;
    unsigned long long *d_diag = nullptr;
    MXA_HIP(hipMalloc(reinterpret_cast<void **>(&d_diag), sizeof(unsigned long long) * 5 * nunits));
    MXA_HIP(hipMemsetAsync(d_diag, 0, sizeof(unsigned long long) * 5 * nunits, s));
    hipLaunchKernelGGL((k_gemm<A, C, MODE, true>), dim3((unsigned)grid), dim3(256), Cfg::kLds, s, G.d, G.pitch, dBp, p.n_pad / 4, dP,
                       p.m_pad, p.n_pad, p.rowblocks, p.nchunks, p.slabs_total, ks, xcd_order, d_diag, split_begin, (const int *)nullptr, (int)nunits, d_ctr, g8, p_split0);
    MXA_HIP(hipStreamSynchronize(s));
    std::vector<unsigned long long> h(5 * nunits);
    MXA_HIP(hipMemcpy(h.data(), d_diag, sizeof(unsigned long long) * 5 * nunits, hipMemcpyDeviceToHost));
    std::vector<double> ghz, cyc;
    for (long i = 0; i < nunits; i++) if (h[2 * i + 1]) {
      const int sp = (int)(i < (long)g8 * p.rowblocks ? ((i & 7) + 8 * ((i >> 3) / p.rowblocks)) / p.nchunks : (i / p.rowblocks) / p.nchunks);
      const int len = ksplit_len(ks, sp);
      ghz.push_back((double)h[2 * i] / (double)h[2 * i + 1] * 0.1); cyc.push_back((double)h[2 * i] / len);
    }
    std::sort(ghz.begin(), ghz.end()); std::sort(cyc.begin(), cyc.end());
    if (!ghz.empty()) printf("MXA_DIAG k_gemm<%d,%d,%d>: %ld pieces on %ld workgroups (splits %d = %d x %d + short x %d), in-kernel clock median %.3f GHz (min %.3f max %.3f); shader cycles per slab median %.0f (ideal %d)\n",
                             A, C, MODE, nunits, grid, p.splits, p.s1, p.l1, p.l2, ghz[ghz.size() / 2], ghz.front(), ghz.back(), cyc[cyc.size() / 2], kSlabSteps * A * C * 16 * 2);
    double ticks = 0.0;   // 100 MHz ticks spent inside the K loops, summed over the pieces
    for (long i = 0; i < nunits; i++) ticks += (double)h[2 * i + 1];
    printf("MXA_DIAG k_gemm: K loops of all pieces = %.3f ms if packed perfectly on %ld resident slots (compare with the kernel duration)\n", ticks * 1e-5 / (double)grid, grid);
    {   // piece lifetimes (100 MHz real-time counter): begin -> end, and where the resident slots stand idle
      double life = 0; unsigned long long first = ~0ull, last = 0;
      for (long i = 0; i < nunits; i++) {
        const unsigned long long b = h[2 * nunits + 2 * i], e = h[4 * nunits + i];
        life += (double)(e - b);
        first = std::min(first, b); last = std::max(last, e);
      }
      printf("MXA_DIAG k_gemm: mean piece lifetime %.1f us; span first begin -> last end %.3f ms; sum of lifetimes / slots = %.3f ms\n",
             life / nunits * 1e-2, (double)(last - first) * 1e-5, life * 1e-5 / (double)grid);
      std::vector<std::pair<unsigned long long, int>> ev;
      ev.reserve(2 * nunits);
      for (long i = 0; i < nunits; i++) { ev.emplace_back(h[2 * nunits + 2 * i], +1); ev.emplace_back(h[4 * nunits + i], -1); }
      std::sort(ev.begin(), ev.end());
      const unsigned long long win = 200000;   // 2 ms in 100 MHz ticks
      double idle[3] = {0, 0, 0}; long active = 0, max_active = 0; unsigned long long prev_t = first;
      for (auto &e : ev) {
        unsigned long long t0 = prev_t, t1 = e.first;
        while (t0 < t1) {   // split the interval at the window borders
          const int zone = t0 < first + win ? 0 : (t0 >= last - win ? 2 : 1);
          const unsigned long long border = zone == 0 ? first + win : (zone == 1 ? last - win : last);
          const unsigned long long te = std::min(t1, std::max(border, t0 + 1));
          idle[zone] += (double)(te - t0) * (double)(grid - std::min(active, grid));
          t0 = te;
        }
        active += e.second; max_active = std::max(max_active, active); prev_t = e.first;
      }
      printf("MXA_DIAG k_gemm: idle slot-time first 2 ms %.3f ms, middle %.3f ms, last 2 ms %.3f ms; max pieces in flight %ld\n",
             idle[0] * 1e-5 / (double)grid, idle[1] * 1e-5 / (double)grid, idle[2] * 1e-5 / (double)grid, max_active);
    }
    (void)hipFree(d_diag);
    return 0;
  }
  hipLaunchKernelGGL((k_gemm<A, C, MODE, false, TR>), dim3((unsigned)grid), dim3(256), Cfg::kLds, s, G.d, G.pitch, dBp, p.n_pad / 4, dP,
                     p.m_pad, p.n_pad, p.rowblocks, p.nchunks, p.slabs_total, ks, xcd_order, (unsigned long long *)nullptr, split_begin, run_if_set, (int)nunits, d_ctr, g8, p_split0);
  MXA_HIP(hipGetLastError());
  return 0;
}

int launch_gemm(const PackedMatrix &G, const double *dBp, double *dP, const GemmPlan &p, int mode, hipStream_t s, int *d_ctr, int split_begin, int split_end,
                const int *run_if_set, bool tr, int p_split0) {
  if (split_end < 0) split_end = p.splits;
  if (p_split0 < 0 || p_split0 > split_begin) { set_error(4, "internal: k_gemm partial-sum base split %d outside [0, %d]", p_split0, split_begin); return 1; }
  if (!G.d) { set_error(4, "internal: k_gemm was given a packed matrix that is not stored (single-orientation object)"); return 1; }
  if (tr) {   // transposed operand: output rows = columns of G (its k individuals), K = rows of G in slabs of 128
    if ((long)p.slabs_total * kSlabK > G.rows_pad || G.nslabs < 1 || (mode != 0 && mode != 3)) {
      set_error(4, "internal: transposed launch does not fit the packed matrix (K slabs %d x 128 > %ld rows) or mode %d", p.slabs_total, G.rows_pad, mode);
      return 1;
    }
#define MXA_DISPATCH_TR(AA, CC)                                                                                          \
    if (p.a == AA && p.c == CC) {                                                                                        \
      if (mode == 3) return launch_gemm_t<AA, CC, 3, true>(G, dBp, dP, p, s, split_begin, split_end, nullptr, d_ctr, p_split0);     \
      return launch_gemm_t<AA, CC, 0, true>(G, dBp, dP, p, s, split_begin, split_end, run_if_set, d_ctr, p_split0);                 \
    }
    MXA_DISPATCH_TR(16, 1) MXA_DISPATCH_TR(16, 2) MXA_DISPATCH_TR(16, 3) MXA_DISPATCH_TR(16, 4)
    MXA_DISPATCH_TR(8, 5) MXA_DISPATCH_TR(8, 6) MXA_DISPATCH_TR(8, 7) MXA_DISPATCH_TR(8, 8)
#undef MXA_DISPATCH_TR
    set_error(5, "internal: no transposed kernel for tile a=%d c=%d", p.a, p.c);
    return 1;
  }
  // host-side shape checks: the kernel reads rows [0, m_pad) x [0, slabs_total*32) bytes and Bp[(k_pad/16)][H][64]
  if (p.m_pad > G.rows_pad || (size_t)p.slabs_total * kSlabBytes > G.pitch) {
    set_error(4, "internal: packed matrix smaller than the launch plan (m_pad %ld > %ld or k bytes %ld > pitch %zu)", p.m_pad,
              G.rows_pad, (long)p.slabs_total * kSlabBytes, G.pitch);
    return 1;
  }
#define MXA_DISPATCH(AA, CC)                                              \
  if (p.a == AA && p.c == CC) {                                           \
    if (mode == 3) return launch_gemm_t<AA, CC, 3>(G, dBp, dP, p, s, split_begin, split_end, nullptr, d_ctr, p_split0);     \
    return launch_gemm_t<AA, CC, 0>(G, dBp, dP, p, s, split_begin, split_end, run_if_set, d_ctr, p_split0);                 \
  }
  MXA_DISPATCH(16, 1)
  MXA_DISPATCH(16, 2)
  MXA_DISPATCH(16, 3)
  MXA_DISPATCH(16, 4)
  MXA_DISPATCH(8, 5)
  MXA_DISPATCH(8, 6)
  MXA_DISPATCH(8, 7)
  MXA_DISPATCH(8, 8)
#undef MXA_DISPATCH
  set_error(5, "internal: no kernel for tile a=%d c=%d", p.a, p.c);
  return 1;
}

// =====================================================================================================
// small-n path (n <= 4): pair-table lookup-add, HBM-bound
// =====================================================================================================
// For n = 1 (the CG / GBLUP iteration of config 5) the MFMA tile wastes 7 of 8 columns and sits 5x off the HBM roofline
// (8 flop per packed byte).  This kernel follows the idea of the reference's CPU "5codes" engine instead
// (src/miraculix/5codesIntern.h:130-266: a table of partial dot products per group of genotypes, one lookup + one add per
// group) in a GPU shape: per slab of KS genotypes the workgroup builds, in LDS, one 16-entry table per genotype PAIR
// (entry v = z0(v)*b[2p] + z1(v)*b[2p+1], one rounding) and every lane walks ITS OWN ROW's nibbles: all 64 lanes of a wave
// look into the same 128-byte table, whose 9 reachable entries sit in distinct banks, so every ds_read is conflict-free.
// Per pair and lane: v_bfe_u32 + v_lshlrev (index), ds_read, NV x v_add_f64.  Rows arrive by LDS-DMA with lane <-> row in
// a chunk-major image [16-byte chunk][row], so the row read is a conflict-free ds_read_b128.
template <int NV, int KS, int WV>
struct LutCfg {
  static constexpr int kThreads = 64 * WV;
  static constexpr int kRows = kThreads;                  // rows per workgroup, one per lane
  static constexpr int kChunks = KS / 64;                 // 16-byte chunks per row per slab
  static constexpr int kRowBytes = kRows * KS / 4;        // packed rows of one slab
  static constexpr int kTabBytes = (KS / 2) * 16 * 8 * NV;
  static constexpr int kBufBytes = kRowBytes + kTabBytes;
  static constexpr int kLds = 2 * kBufBytes;
};

template <int NV, int KS, int WV>
__global__ void __launch_bounds__(64 * WV)
k_lut(const uint8_t *__restrict__ G, size_t pitch, const double *__restrict__ B, long ldb, long k, int n, double *__restrict__ P,
      long m_pad, int rowblocks, int slabs_total, int slabs_per_split, const int *__restrict__ run_if_set) {
  using Cfg = LutCfg<NV, KS, WV>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  if (run_if_set && *run_if_set == 0) return;   // fallback launch of the guarded small-n route: the exact int8 chain is doing this product
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int rb = blockIdx.x % rowblocks, sp = blockIdx.x / rowblocks;
  const int slab0 = sp * slabs_per_split, slab1 = min(slab0 + slabs_per_split, slabs_total);
  const long row0 = (long)rb * Cfg::kRows;
  const uint32_t lds0 = (uint32_t)(size_t)(lptr_t)smem;
  // tiled layout: tile (row0/256 + wave/4, slab) holds 256 rows x 32 B contiguously; lane <-> row, one 16-byte chunk per
  // instruction, so the LDS image is chunk-major [chunk][row] (conflict-free ds_read_b128 per lane)
  static_assert(KS == kSlabK, "the lookup kernel walks the staged slabs");
  const uint32_t r_lane = (uint32_t)lane * kSlabBytes;
  const char *G_u = reinterpret_cast<const char *>(G) + (size_t)(row0 / kTileRows + (wave >> 2)) * (pitch / kSlabBytes) * kTileBytes +
                    (size_t)((wave & 3) * 64) * kSlabBytes;

  auto issue_rows = [&](int slab, int buf) {
    const char *src = G_u + (size_t)slab * kTileBytes;
#pragma unroll
    for (int c = 0; c < Cfg::kChunks; c++) dma16_p<MXA_NT_LUT != 0>(src + c * 16, r_lane, lds0 + buf * Cfg::kBufBytes + (c * Cfg::kRows + wave * 64) * 16);
  };
  // Table build: the slab has KS/2 pair tables x 16 entries; thread t writes entries q = t + T*e (pair q>>4, nibble q&15), so a
  // wave writes 64 consecutive entries (conflict-free ds_write).  The pair's two B values are loaded one slab ahead into
  // registers (global latency hides behind a slab of lookups), the table is written one slab ahead of its use.
  constexpr int kEnt = (KS / 2) * 16 / Cfg::kThreads;       // entries per thread per slab
  static_assert((KS / 2) * 16 % Cfg::kThreads == 0, "whole entries per thread");
  auto load_b = [&](int slab, double (&b0)[kEnt][NV], double (&b1)[kEnt][NV]) {
#pragma unroll
    for (int e = 0; e < kEnt; e++) {
      const long kk = (long)slab * KS + 2 * ((tid + Cfg::kThreads * e) >> 4);
#pragma unroll
      for (int j = 0; j < NV; j++) {
        // unconditional loads from clamped addresses, then a select: a branch around each load would serialise them
        // (every load followed by its own vmcnt(0))
        const long jc = j < n ? j : 0;
        const double v0 = B[(kk < k ? kk : k - 1) + jc * ldb];
        const double v1 = B[(kk + 1 < k ? kk + 1 : k - 1) + jc * ldb];
        b0[e][j] = (j < n && kk < k) ? v0 : 0.0;
        b1[e][j] = (j < n && kk + 1 < k) ? v1 : 0.0;
      }
    }
  };
  auto build = [&](int buf, const double (&b0)[kEnt][NV], const double (&b1)[kEnt][NV]) {
    double *tab = reinterpret_cast<double *>(smem + buf * Cfg::kBufBytes + Cfg::kRowBytes);
    const double z0 = (double)(tid & 3), z1 = (double)((tid >> 2) & 3);
#pragma unroll
    for (int e = 0; e < kEnt; e++)
#pragma unroll
      for (int j = 0; j < NV; j++) tab[(size_t)(tid + Cfg::kThreads * e) * NV + j] = fma(z1, b1[e][j], z0 * b0[e][j]);
  };

  double acc[4][NV];
#pragma unroll
  for (int a = 0; a < 4; a++)
#pragma unroll
    for (int j = 0; j < NV; j++) acc[a][j] = 0.0;

  double bc0[kEnt][NV], bc1[kEnt][NV], bn0[kEnt][NV], bn1[kEnt][NV];
  if (slab0 < slab1) {
    issue_rows(slab0, 0);
    load_b(slab0, bc0, bc1);
    build(0, bc0, bc1);
    if (slab0 + 1 < slab1) load_b(slab0 + 1, bc0, bc1);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int slab = slab0; slab < slab1; slab++) {
    const int buf = (slab - slab0) & 1;
    if (slab + 1 < slab1) {
      issue_rows(slab + 1, buf ^ 1);
      if (slab + 2 < slab1) load_b(slab + 2, bn0, bn1);
      build(buf ^ 1, bc0, bc1);
#pragma unroll
      for (int e = 0; e < kEnt; e++)
#pragma unroll
        for (int j = 0; j < NV; j++) { bc0[e][j] = bn0[e][j]; bc1[e][j] = bn1[e][j]; }
    }
    const char *rows = smem + buf * Cfg::kBufBytes + tid * 16;
    // LDS byte address of the table region of this buffer (128-byte aligned, so the entry index can be OR-ed in)
    const uint32_t tab = lds0 + buf * Cfg::kBufBytes + Cfg::kRowBytes;
#pragma unroll
    for (int c = 0; c < Cfg::kChunks; c++) {
      const uint4 w4 = *reinterpret_cast<const uint4 *>(rows + c * Cfg::kRows * 16);
#pragma unroll
      for (int d = 0; d < 4; d++) {
        const uint32_t w = d == 0 ? w4.x : d == 1 ? w4.y : d == 2 ? w4.z : w4.w;
#pragma unroll
        for (int q = 0; q < 8; q++) {
          // v_bfe_u32 + v_lshl_or_b32: address = table region | (nibble * entry size); the pair's table is the immediate offset
          const uint32_t idx = __builtin_amdgcn_ubfe(w, 4 * q, 4);
          uint32_t addr;   // asm keeps the 2-instruction form (hipcc otherwise rewrites it into shift + and + or)
          if (NV == 1) asm("v_lshl_or_b32 %0, %1, 3, %2" : "=v"(addr) : "v"(idx), "v"(tab));
          else if (NV == 2) asm("v_lshl_or_b32 %0, %1, 4, %2" : "=v"(addr) : "v"(idx), "v"(tab));
          else asm("v_lshl_or_b32 %0, %1, 5, %2" : "=v"(addr) : "v"(idx), "v"(tab));
          __builtin_assume((addr & (8 * NV - 1)) == 0);   // the asm hides the alignment: without this the read is split into dwords
          using lds_cd = const __attribute__((address_space(3))) double;
          lds_cd *e = (lds_cd *)(size_t)(addr + (c * 32 + d * 8 + q) * 16 * 8 * NV);
#pragma unroll
          for (int j = 0; j < NV; j++) acc[q & 3][j] += e[j];
        }
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  double *Pbase = P + (size_t)sp * NV * m_pad;
#pragma unroll
  for (int j = 0; j < NV; j++) Pbase[(size_t)j * m_pad + row0 + tid] = (acc[0][j] + acc[1][j]) + (acc[2][j] + acc[3][j]);
}

constexpr int kLutWaves = 8;   // 512 lanes = 512 rows per workgroup share one set of tables

template <int NV, int KS>
static int launch_lut_t(const PackedMatrix &G, const double *dB, long ldb, int n, double *dP, const GemmPlan &p, hipStream_t s, const int *run_if_set) {
  using Cfg = LutCfg<NV, KS, kLutWaves>;
  static unsigned long long attr_mask = 0;
  if (ensure_dyn_lds(reinterpret_cast<const void *>(&k_lut<NV, KS, kLutWaves>), Cfg::kLds, &attr_mask)) return 1;
  const long grid = (long)p.rowblocks * p.splits;
  hipLaunchKernelGGL((k_lut<NV, KS, kLutWaves>), dim3((unsigned)grid), dim3(Cfg::kThreads), Cfg::kLds, s, G.d, G.pitch, dB, ldb, G.k, n, dP, p.m_pad,
                     p.rowblocks, p.slabs_total, p.slabs_per_split, run_if_set);
  MXA_HIP(hipGetLastError());
  return 0;
}

constexpr int kLutKS = 128;   // genotypes per slab of the lookup kernel (= kSlabK, so the staged pitch fits)

GemmPlan plan_lut(long m, long k_pad, int n) { return plan_lut_host(m, k_pad, n); }

static_assert(kPlanWaves == kWaves && kPlanSlabSteps == kSlabSteps && kPlanSlabK == kSlabK && kPlanLutWaves == kLutWaves && kPlanLutKS == kLutKS,
              "mxa_plan.h restates the kernels' geometry for the host-only planners");

int launch_lut(const PackedMatrix &G, const double *dB, long ldb, int n, double *dP, const GemmPlan &p, hipStream_t s, const int *run_if_set) {
  if (!G.d) { set_error(4, "internal: k_lut was given a packed matrix that is not stored (single-orientation object)"); return 1; }
  if (p.m_pad > G.rows_pad || (size_t)p.slabs_total * (kLutKS / 4) > G.pitch) { set_error(4, "internal: packed matrix smaller than the lookup plan"); return 1; }
  if (p.n_pad == 1) return launch_lut_t<1, kLutKS>(G, dB, ldb, n, dP, p, s, run_if_set);
  if (p.n_pad == 2) return launch_lut_t<2, kLutKS>(G, dB, ldb, n, dP, p, s, run_if_set);
  return launch_lut_t<4, kLutKS>(G, dB, ldb, n, dP, p, s, run_if_set);
}

// =====================================================================================================
// finish: split-K reduction (ascending split order) + centring + ldc store
// =====================================================================================================
// 'N' (mode_trans=0): C[i,j] = sum_s P + (-2 * sum_k f_k B[k,j])            (x=f, y=1: dgemm_compressed_cuda.cu:426-459)
// 'T' (mode_trans=1): C[s,j] = sum_s P + (-2 * sum_i B[i,j]) * f_s           (x=1, y=f)
// rows m..fill_rows-1 of every column are zero-filled (plain ABI: fill_rows = ldc, like the reference CPU path, 5codesIntern.h:67).
// Round 6: two consecutive rows per thread (16-byte loads of P; a 16-byte store where C's column is 16-byte aligned) and the tile index by shift where the
// tile height is a power of two (k_gemm's row blocks: 128 / 256; a 64-bit division per element cost as much as the memory traffic: 220 us for the 512 MB
// of a C2 product: 184 us now).  Same additions in the same order: bit-identical.  (A block per partial-sum TILE walking the columns -- P read front to
// back -- was measured too: 258 us; the scattered 1 KiB runs of C cost more than the strided reads of P.)
__global__ void __launch_bounds__(256) k_finish(const double *__restrict__ P, long m_pad, long p_rows, int p_shift, int n_pad, int splits, long m, int n,
                                                double *__restrict__ Cout, long ldc, long fill_rows, int mode_trans, int centered,
                                                const double *__restrict__ sumB, const double *__restrict__ sumfB,
                                                const double *__restrict__ f, const int *__restrict__ E, int back, int e_splits, int e_stride,
                                                const int *__restrict__ run_if_set, const int *__restrict__ unscaled_if_set, int group) {
  const int j = blockIdx.y;
  const long r0 = 2 * ((long)blockIdx.x * blockDim.x + threadIdx.x);
  if (run_if_set && *run_if_set == 0) return;
  if (unscaled_if_set && *unscaled_if_set != 0) E = nullptr;   // the MODE 0 fallback produced these partial sums: nothing to scale back
  if (r0 >= fill_rows) return;
  // K-split GROUPS (products whose partial sums would not fit the device: gemm_grouped): `splits` partials of one group are added to the running
  // sum kept, unscaled and uncentred, in C itself -- bit 0: start from C instead of 0, bit 1: not the last group (store the raw sum).  The
  // additions run in the same ascending split order as one pass over all splits: bit-identical.
  const bool acc_in = group & 1, raw_out = group & 2;
  const bool two = r0 + 1 < fill_rows;
  double *cp = Cout + r0 + (long)j * ldc;
  double v0 = 0.0, v1 = 0.0;
  if (acc_in) { if (r0 < m) v0 = cp[0]; if (two && r0 + 1 < m) v1 = cp[1]; }
  if (r0 < m) {
    // rows r0, r0 + 1 lie in one tile (r0 even, p_rows even: k_gemm's 128 / 256, the lookup kernel's m_pad -- a multiple of 512)
    const size_t tile = p_shift >= 0 ? (size_t)(r0 >> p_shift) : (size_t)(r0 / p_rows), within = p_shift >= 0 ? (size_t)(r0 & (p_rows - 1)) : (size_t)(r0 % p_rows);
    const size_t ntiles = p_shift >= 0 ? (size_t)(m_pad >> p_shift) : (size_t)(m_pad / p_rows);
    const size_t sstride = ntiles * (size_t)n_pad * (size_t)p_rows;            // doubles between the splits
    const double *pp = P + ((tile * n_pad + j) * (size_t)p_rows + within);
    typedef double v2d __attribute__((ext_vector_type(2)));
    const bool per_split_scale = !raw_out && E && e_splits > 0;
    for (int s = 0; s < splits; s++) {
      const v2d x = __builtin_nontemporal_load(reinterpret_cast<const v2d *>(pp + (size_t)s * sstride));   // read once (the padding row of an odd m is there: m_pad is even)
      if (per_split_scale) {
        // host-operand pipeline: B was scaled per K chunk (e_splits splits share one exponent row of E).  Every partial is scaled back
        // BEFORE it is added, in the same ascending order: power-of-two scaling commutes with rounding, so the sum is bit-identical to the
        // one-exponent path below
        const int ex = back + E[(size_t)(s / e_splits) * e_stride + j];
        v0 += ldexp(x[0], ex); v1 += ldexp(x[1], ex);
      } else { v0 += x[0]; v1 += x[1]; }
    }
    if (!raw_out) {
      if (E && !per_split_scale) { const int ex = back + E[j]; v0 = ldexp(v0, ex); v1 = ldexp(v1, ex); }   // undo the operand scaling of the denormal-operand mode (exact)
      if (centered) {
        if (mode_trans) { const double c = -2.0 * sumB[j]; v0 = fma(c, f[r0], v0); if (r0 + 1 < m) v1 = fma(c, f[r0 + 1], v1); }
        else { const double c = -2.0 * sumfB[j]; v0 += c; v1 += c; }
      }
    }
    if (r0 + 1 >= m) v1 = 0.0;                                                  // row m (ld padding of the plain ABI) is zero
  }
  if (two && (reinterpret_cast<size_t>(cp) & 15) == 0) {
    typedef double v2d __attribute__((ext_vector_type(2)));
    *reinterpret_cast<v2d *>(cp) = v2d{v0, v1};
  } else { cp[0] = v0; if (two) cp[1] = v1; }
}

int launch_finish(const double *dP, const GemmPlan &p, long m, int n, double *dC, long ldc, long fill_rows, int mode_trans, bool centered,
                  const double *d_sumB, const double *d_sumfB, const double *d_f, hipStream_t s, const int *d_E, int e_splits, int e_stride, const int *run_if_set,
                  const int *unscaled_if_set, int group_splits, int group) {
  if ((p.p_rows & 1) || (p.m_pad & 1)) { set_error(4, "internal: k_finish wants even tile heights (p_rows %ld, m_pad %ld)", p.p_rows, p.m_pad); return 1; }
  int p_shift = -1;
  if ((p.p_rows & (p.p_rows - 1)) == 0) { p_shift = 0; while ((1L << p_shift) < p.p_rows) p_shift++; }
  dim3 grid((unsigned)((fill_rows + 511) / 512), n);
  hipLaunchKernelGGL(k_finish, grid, dim3(256), 0, s, dP, p.m_pad, p.p_rows, p_shift, p.n_pad, group_splits > 0 ? group_splits : p.splits, m, n, dC, ldc, fill_rows, mode_trans,
                     centered ? 1 : 0, d_sumB, d_sumfB, d_f, d_E, 1074 - kDenUp, e_splits, e_stride, run_if_set, unscaled_if_set, group);
  MXA_HIP(hipGetLastError());
  return 0;
}


// sum of per-shard partial results in ascending shard order (fixed order: bitwise reproducible), ld-padded store
__global__ void __launch_bounds__(256) k_reduce_parts(PartList parts, long m, double *__restrict__ Cout, long ldc, long fill_rows) {
  const int j = blockIdx.y;
  const long r = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= fill_rows) return;
  double v = 0.0;
  if (r < m) {
    v = parts.p[0][r + (size_t)j * m];
    for (int g = 1; g < parts.count; g++) v += parts.p[g][r + (size_t)j * m];
  }
  Cout[r + (size_t)j * ldc] = v;
}

int launch_reduce_parts(const PartList &parts, long m, int n, double *dC, long ldc, long fill_rows, hipStream_t s) {
  if (n <= 0 || parts.count <= 0) return 0;
  dim3 grid((unsigned)((fill_rows + 255) / 256), n);
  hipLaunchKernelGGL(k_reduce_parts, grid, dim3(256), 0, s, parts, m, dC, ldc, fill_rows);
  MXA_HIP(hipGetLastError());
  return 0;
}

int ensure_dyn_lds(const void *func, int bytes, unsigned long long *mask) {
  int dev = 0;
  MXA_HIP(hipGetDevice(&dev));
  const unsigned long long bit = 1ull << (dev & 63);
  if (__atomic_load_n(mask, __ATOMIC_ACQUIRE) & bit) return 0;
  MXA_HIP(hipFuncSetAttribute(func, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
  __atomic_fetch_or(mask, bit, __ATOMIC_RELEASE);
  return 0;
}

// =====================================================================================================
// staging helpers: 2-bit transpose, allele frequencies (raw PLINK codes in, raw codes out)
// =====================================================================================================
// 64 x 64 genotype tiles through LDS.  in: `rows` rows of ceil(cols/4) bytes; out: `cols` rows of ceil(rows/4) bytes.
// ALIGNED: both row pitches are multiples of 4 bytes -> dword loads and stores (4x fewer memory instructions).
template <bool ALIGNED>
__global__ void __launch_bounds__(256) k_transpose_2bit(const uint8_t *__restrict__ in, long rows, long cols,
                                                        uint8_t *__restrict__ out, long nbx, long blk0) {
  __shared__ uint8_t tile[64][20];
  const long bin = (cols + 3) / 4, bout = (rows + 3) / 4;
  // 1-D grid (either dimension may exceed the 65535 limit of gridDim.y): block = by * nbx + bx; blk0 = first block of this launch
  const long bid = blk0 + blockIdx.x;
  const long r0 = (bid / nbx) * 64, c0 = (bid % nbx) * 64;
  {
    const int r = threadIdx.x >> 2, part = threadIdx.x & 3;
    const long byte0 = c0 / 4 + part * 4;
    uint32_t w = 0;
    if (r0 + r < rows) {
      if (ALIGNED && byte0 + 3 < bin) {
        w = *reinterpret_cast<const uint32_t *>(in + (size_t)(r0 + r) * bin + byte0);
      } else {
        for (int u = 0; u < 4; u++)
          if (byte0 + u < bin) w |= (uint32_t)in[(size_t)(r0 + r) * bin + byte0 + u] << (8 * u);
      }
    }
    *reinterpret_cast<uint32_t *>(&tile[r][part * 4]) = w;
  }
  __syncthreads();
  {
    const int c = threadIdx.x >> 2, q = threadIdx.x & 3;  // output row c0+c, output bytes 4q..4q+3 of this tile
    if (c0 + c < cols) {
      const long obyte0 = r0 / 4 + 4 * q;
      uint32_t wout = 0;
      for (int ob = 0; ob < 4; ob++) {
        uint32_t v = 0;
        for (int u = 0; u < 4; u++) {
          const int r = 16 * q + 4 * ob + u;
          uint32_t code = (tile[r][c >> 2] >> (2 * (c & 3))) & 3u;
          if (r0 + r >= rows) code = 0;
          v |= code << (2 * u);
        }
        wout |= v << (8 * ob);
      }
      if (ALIGNED && obyte0 + 3 < bout) {
        *reinterpret_cast<uint32_t *>(out + (size_t)(c0 + c) * bout + obyte0) = wout;
      } else {
        for (int ob = 0; ob < 4; ob++)
          if (obyte0 + ob < bout) out[(size_t)(c0 + c) * bout + obyte0 + ob] = (uint8_t)(wout >> (8 * ob));
      }
    }
  }
}

// Fast path (both row pitches multiples of 4 bytes): 256 rows x 512 columns per workgroup.  Full 128-byte runs of 256 input rows
// go to LDS; every thread transposes two 16 x 16 blocks of 2-bit fields in registers with a 4-stage butterfly (masked swaps at
// distances 8, 4, 2, 1 -- 192 integer ops per 256 genotypes, no per-field gather); the transposed dwords go back through the same
// LDS buffer so that the 512 output rows leave as 64-byte runs.
__device__ __forceinline__ void swap_fields(uint32_t &lo, uint32_t &hi, int sh, uint32_t mask) {
  const uint32_t t = ((lo >> sh) ^ hi) & mask;   // exchange the high half-blocks of lo with the low half-blocks of hi
  hi ^= t;
  lo ^= t << sh;
}
__device__ __forceinline__ void transpose16x16_2bit(uint32_t (&a)[16]) {
#pragma unroll
  for (int i = 0; i < 8; i++) swap_fields(a[i], a[i + 8], 16, 0x0000FFFFu);
#pragma unroll
  for (int i = 0; i < 16; i++) if (!(i & 4)) swap_fields(a[i], a[i + 4], 8, 0x00FF00FFu);
#pragma unroll
  for (int i = 0; i < 16; i++) if (!(i & 2)) swap_fields(a[i], a[i + 2], 4, 0x0F0F0F0Fu);
#pragma unroll
  for (int i = 0; i < 16; i++) if (!(i & 1)) swap_fields(a[i], a[i + 1], 2, 0x33333333u);
}

struct __attribute__((packed, aligned(4))) U4 { uint32_t x[4]; };   // 16 bytes with 4-byte alignment: global_load/store_dwordx4
constexpr int kTrRows = 256, kTrCols = 512;            // genotypes per tile
constexpr int kTrInPitch = kTrCols / 16 + 1;           // dwords per tile row in LDS (+1: conflict-free column reads)
constexpr int kTrOutPitch = kTrRows / 16 + 1;

__global__ void __launch_bounds__(256) k_transpose_2bit_tiled(const uint8_t *__restrict__ in, long rows, long cols, uint8_t *__restrict__ out, long nbx, long blk0) {
  __shared__ uint32_t lds[(kTrRows * kTrInPitch > kTrCols * kTrOutPitch) ? kTrRows * kTrInPitch : kTrCols * kTrOutPitch];
  const long bin = (cols + 3) / 4, bout = (rows + 3) / 4;   // multiples of 4 (checked by the launcher)
  const long bid = blk0 + blockIdx.x;
  const long r0 = (bid / nbx) * kTrRows, c0 = (bid % nbx) * kTrCols;
  const long din = bin / 4, dout = bout / 4;                // dwords per row
  const uint32_t *in32 = reinterpret_cast<const uint32_t *>(in);
  uint32_t *out32 = reinterpret_cast<uint32_t *>(out);
  // phase 1: 256 rows x 32 dwords, 8 threads x 4 dwords per row
#pragma unroll
  for (int it = 0; it < 8; it++) {
    const int idx = threadIdx.x + 256 * it;
    const int r = idx >> 3, part = idx & 7;
    const long gr = r0 + r, gd = c0 / 16 + part * 4;
    uint32_t w[4] = {0u, 0u, 0u, 0u};
    if (gr < rows) {
      const uint32_t *p = in32 + (size_t)gr * din + gd;
      if (gd + 3 < din) { const U4 v = *reinterpret_cast<const U4 *>(p); w[0] = v.x[0]; w[1] = v.x[1]; w[2] = v.x[2]; w[3] = v.x[3]; }   // one 16-byte load, 4-byte aligned
      else { for (int u = 0; u < 4; u++) if (gd + u < din) w[u] = p[u]; }
    }
#pragma unroll
    for (int u = 0; u < 4; u++) lds[r * kTrInPitch + part * 4 + u] = w[u];
  }
  __syncthreads();
  // phase 2: block (br, bc) = rows 16 br .. +15, dword column bc
  uint32_t a[2][16];
#pragma unroll
  for (int pass = 0; pass < 2; pass++) {
    const int bc = threadIdx.x & 31, br = (threadIdx.x >> 5) + 8 * pass;
#pragma unroll
    for (int i = 0; i < 16; i++) a[pass][i] = lds[(16 * br + i) * kTrInPitch + bc];
    transpose16x16_2bit(a[pass]);
  }
  __syncthreads();
  // phase 3: a[f] is dword br of output row 16 bc + f
#pragma unroll
  for (int pass = 0; pass < 2; pass++) {
    const int bc = threadIdx.x & 31, br = (threadIdx.x >> 5) + 8 * pass;
#pragma unroll
    for (int f = 0; f < 16; f++) lds[(16 * bc + f) * kTrOutPitch + br] = a[pass][f];
  }
  __syncthreads();
  // phase 4: 512 output rows x 16 dwords, 4 threads x 4 dwords per row
#pragma unroll
  for (int it = 0; it < 8; it++) {
    const int idx = threadIdx.x + 256 * it;
    const int orow = idx >> 2, part = idx & 3;
    const long gc = c0 + orow, gd = r0 / 16 + part * 4;
    if (gc < cols) {
      uint32_t *p = out32 + (size_t)gc * dout + gd;
      U4 v;
#pragma unroll
      for (int u = 0; u < 4; u++) v.x[u] = lds[orow * kTrOutPitch + part * 4 + u];
      if (gd + 3 < dout) *reinterpret_cast<U4 *>(p) = v;
      else { for (int u = 0; u < 4; u++) if (gd + u < dout) p[u] = v.x[u]; }
    }
  }
}

int launch_transpose_2bit(const uint8_t *d_in, long rows, long cols, uint8_t *d_out, hipStream_t s) {
  if (rows <= 0 || cols <= 0) return 0;
  const long bin = (cols + 3) / 4, bout = (rows + 3) / 4;
  const bool aligned = (bin % 4 == 0) && (bout % 4 == 0) && (reinterpret_cast<uintptr_t>(d_in) % 4 == 0) && (reinterpret_cast<uintptr_t>(d_out) % 4 == 0);
  constexpr bool no_tiled = false;
  // a launch may not exceed 2^32 threads: the grid is issued in chunks of 2^23 blocks (found by the 625k x 200k full-size test:
  // 30.5 M blocks x 256 threads were silently truncated to the grid modulo 2^24)
  if (aligned && !no_tiled) {
    // fields beyond `rows` / `cols` inside the last dwords are zero in the input (PLINK padding) and come out as zero padding
    const long nbx = (cols + kTrCols - 1) / kTrCols, nby = (rows + kTrRows - 1) / kTrRows;
    for (long b0 = 0; b0 < nbx * nby; b0 += kMaxBlocksPerLaunch)
      hipLaunchKernelGGL(k_transpose_2bit_tiled, dim3((unsigned)std::min(kMaxBlocksPerLaunch, nbx * nby - b0)), dim3(256), 0, s, d_in, rows, cols, d_out, nbx, b0);
    MXA_HIP(hipGetLastError());
    return 0;
  }
  const long nbx = (cols + 63) / 64, nby = (rows + 63) / 64;
  for (long b0 = 0; b0 < nbx * nby; b0 += kMaxBlocksPerLaunch) {
    const unsigned nb = (unsigned)std::min(kMaxBlocksPerLaunch, nbx * nby - b0);
    if (aligned) hipLaunchKernelGGL(k_transpose_2bit<true>, dim3(nb), dim3(256), 0, s, d_in, rows, cols, d_out, nbx, b0);
    else hipLaunchKernelGGL(k_transpose_2bit<false>, dim3(nb), dim3(256), 0, s, d_in, rows, cols, d_out, nbx, b0);
  }
  MXA_HIP(hipGetLastError());
  return 0;
}

// f_s = (sum of allele counts of SNP s, missing counted 0) / (2*indiv); one wave per SNP row, 16 bytes per lane per step when
// the rows are 16-byte aligned (then the tail bytes, byte-wise)
__device__ __forceinline__ unsigned count16(uint32_t w) {   // sum of the 16 allele counts in a dword of PLINK codes
  const uint32_t z = recode16(w);
  const uint32_t s2 = (z & 0x33333333u) + ((z >> 2) & 0x33333333u);      // 8 nibbles, each <= 4
  const uint32_t s4 = (s2 & 0x0F0F0F0Fu) + ((s2 >> 4) & 0x0F0F0F0Fu);    // 4 bytes, each <= 8
  return (s4 * 0x01010101u) >> 24;
}

__global__ void __launch_bounds__(256) k_allele_freq(const uint8_t *__restrict__ plink, long snps, long indiv, double *__restrict__ f) {
  const long s = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (s >= snps) return;
  const long bps = (indiv + 3) / 4;
  const uint8_t *row = plink + (size_t)s * bps;
  unsigned long long cnt = 0;
  const long full = indiv / 4;                              // bytes whose 4 fields all exist
  // head bytes up to the first 16-byte boundary, then 16 bytes per lane per step, then the tail
  long head = (long)((16 - (reinterpret_cast<uintptr_t>(row) & 15)) & 15);
  if (head > full) head = full;
  const long nvec = (full - head) / 16;
  for (long v = lane; v < nvec; v += 64) {
    const uint4 w = *reinterpret_cast<const uint4 *>(row + head + v * 16);
    cnt += count16(w.x) + count16(w.y) + count16(w.z) + count16(w.w);
  }
  for (long b = lane; b < head; b += 64) cnt += count16(row[b]);
  for (long b = head + nvec * 16 + lane; b < bps; b += 64) {
    uint32_t w = row[b];
    long valid = indiv - 4 * b;
    if (valid < 4) w &= (1u << (2 * valid)) - 1u;
    cnt += count16(w);
  }
  for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o);
  if (lane == 0) f[s] = (double)cnt / (2.0 * (double)indiv);
}

int launch_allele_freq(const uint8_t *d_plink, long snps, long indiv, double *d_f, hipStream_t s) {
  if (snps <= 0) return 0;
  hipLaunchKernelGGL(k_allele_freq, dim3((unsigned)((snps + 3) / 4)), dim3(256), 0, s, d_plink, snps, indiv, d_f);
  MXA_HIP(hipGetLastError());
  return 0;
}

}  // namespace mxa
