// mxa_sparse.hip -- sparse_times_plink: C (nIdx x entries) = S (CSR, nIdx x rows) * unpack(P), P = rows x ceil(entries/4) raw PLINK bytes.
// Replaces sparseTGenoPlink (reference src/miraculix/plinkUint.cc:352-470 behind 5codesAPI.c:135-157).  HBM/L2-bound gather:
// every stored entry of S pulls one packed row through the cache; per packed byte 4 decodes + 4 fp64 FMAs.
//
// Workgroup = 16 sparse rows x 512 entries (128 packed bytes).  Wave w takes sparse rows w, w+4, ...; a lane takes 2 packed bytes
// (8 entries), so one wave-load is a 128-byte run of the packed row; the CSR entries of a row are wave-uniform (scalar loads).
// Sums are accumulated in CSR order -- the order of the reference's loop over the stored entries -- and transposed through LDS so
// that the store to the column-major C runs along the sparse-row index (128-byte runs).
#include "mxa_internal.h"

namespace mxa {

constexpr int kSpJ = 16;      // sparse rows per workgroup
constexpr int kSpE = 512;     // entries per workgroup
constexpr int kSpPitch = kSpE + 1;

__global__ void __launch_bounds__(256) k_sparse_times_plink(const uint8_t *__restrict__ P, size_t pitch, long entries, int nIdx, const int *__restrict__ rowIdx,
                                                            const int *__restrict__ colIdx, const double *__restrict__ val, double *__restrict__ C,
                                                            long ldc, long e_base) {
  __shared__ double tile[kSpJ * kSpPitch];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long e0 = (long)blockIdx.x * kSpE;            // first entry of this block (relative to the slab's e_base)
  const int j0 = blockIdx.y * kSpJ;
  const long byte0 = (e_base + e0) / 4 + 2 * lane;    // e_base and e0 are multiples of 4
  const long nbytes = (entries + 3) / 4;
  for (int jj = wave; jj < kSpJ; jj += 4) {
    const int j = j0 + jj;
    double acc[8];
#pragma unroll
    for (int i = 0; i < 8; i++) acc[i] = 0.0;
    if (j < nIdx) {
      const int b = rowIdx[j], e = rowIdx[j + 1];
      for (int t = b; t < e; t++) {
        const uint8_t *row = P + (size_t)colIdx[t] * pitch;
        const double a = val[t];
        uint32_t w = 0;
        if (byte0 + 1 < nbytes) w = (uint32_t)row[byte0] | ((uint32_t)row[byte0 + 1] << 8);
        else if (byte0 < nbytes) w = row[byte0];
#pragma unroll
        for (int i = 0; i < 8; i++) {
          const uint32_t c = (w >> (2 * i)) & 3u;                    // PLINK: 00 -> 0, 01 (missing) -> 0, 10 -> 1, 11 -> 2
          const double z = (double)((c >> 1) + (c == 3u ? 1u : 0u));
          acc[i] = fma(a, z, acc[i]);
        }
      }
    }
#pragma unroll
    for (int i = 0; i < 8; i++) tile[jj * kSpPitch + 8 * lane + i] = acc[i];
  }
  __syncthreads();
  const int jj = threadIdx.x & 15;
  for (int ee = threadIdx.x >> 4; ee < kSpE; ee += 16) {
    const long eg = e_base + e0 + ee;
    if (eg < entries) {
      const long r = j0 + jj;
      if (r < nIdx) C[r + (e0 + ee) * ldc] = tile[jj * kSpPitch + ee];
    }
  }
}

// zero-fill of the ld padding rows [nIdx, ldc) of a slab (the reference memsets Ldc x entries first: haplogeno.cc:1696)
__global__ void k_zero_ld_rows(double *__restrict__ C, long ldc, int nIdx, long cols) {
  const long pad = ldc - nIdx;
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= pad * cols) return;
  C[nIdx + idx % pad + (idx / pad) * ldc] = 0.0;
}

int launch_sparse_times_plink(const uint8_t *dP, size_t pitch, long entries, int nIdx, const int *d_rowIdx, const int *d_colIdx, const double *d_val,
                              double *dC_slab, long ldc, long e_base, long e_count, hipStream_t s) {
  if (e_count <= 0 || nIdx <= 0) return 0;
  dim3 grid((unsigned)((e_count + kSpE - 1) / kSpE), (unsigned)((nIdx + kSpJ - 1) / kSpJ));
  if (grid.y > 65535) { set_error(8, "sparse_times_plink: more than %d sparse rows are not supported", 65535 * kSpJ); return 1; }
  hipLaunchKernelGGL(k_sparse_times_plink, grid, dim3(256), 0, s, dP, pitch, entries, nIdx, d_rowIdx, d_colIdx, d_val, dC_slab, ldc, e_base);
  if (ldc > nIdx) {
    const long tot = (ldc - nIdx) * e_count;
    hipLaunchKernelGGL(k_zero_ld_rows, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, dC_slab, ldc, nIdx, e_count);
  }
  MXA_HIP(hipGetLastError());
  return 0;
}

}  // namespace mxa
