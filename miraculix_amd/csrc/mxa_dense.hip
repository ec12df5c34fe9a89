// mxa_dense.hip -- dense fp64 building blocks of the solver twin (mxa_solve.hip), hand-written for gfx950 (round 3; round 2 bound
// rocblas_dtrsm / dsyrk / dgemm with dlopen -- 2.4 s on the first call just to load the library).  Not on the compressed-genotype hot
// path (last item); reference counterpart: the cuSOLVER / cuBLAS calls of src/cuda/solve_cuda.cu:70-280.
//
//   k_dgemm<TA, TB>   C = alpha * op(A) op(B) + beta * C on v_mfma_f64_16x16x4_f64, column-major, any sizes and leading dimensions.
//                     Workgroup tile 128 x 128, four waves of 64 x 64 (4 x 4 MFMA tiles: 128 accumulator registers), K in steps of 16
//                     through an LDS stage [k][m] / [k][n] (row pitch 144 doubles: the two k rows a half-wave reads fall into
//                     different bank halves), the next stage's global loads in flight (registers) under the current stage's MFMAs.
//                     `lower_only`: workgroups whose tile lies strictly above the diagonal leave at once (the rank-k update of the
//                     Cholesky trailing matrix).  C may alias A or B when every workgroup reads only what it writes (the in-place
//                     triangular-solve steps below): all of a workgroup's operand reads precede its stores.
//   k_potrf_inv_block Cholesky factor of a 64 x 64 diagonal block and the inverse of that factor, in LDS: the triangular solves (inside the
//                     factorisation and of the right-hand sides) then are products with the inverted blocks.
#include "mxa_internal.h"

namespace mxa {

typedef double double4_t __attribute__((ext_vector_type(4)));

constexpr int kDgBM = 128, kDgBN = 128, kDgBK = 16, kDgPitch = 144;

template <int TA, int TB>
__global__ void __launch_bounds__(256) k_dgemm(long M, long N, long K, double alpha, const double *A, long lda, const double *B, long ldb, double beta, double *C,
                                               long ldc, int lower_only) {
  __shared__ double As[kDgBK][kDgPitch], Bs[kDgBK][kDgPitch];
  const long m0 = (long)blockIdx.x * kDgBM, n0 = (long)blockIdx.y * kDgBN;
  if (lower_only && n0 > m0 + kDgBM - 1) return;
  const int t = threadIdx.x, lane = t & 63, w = t >> 6, wm = w & 1, wn = w >> 1;
  // element (m, k) of op(A): TA == 0: A[m + k lda] (A is M x K); TA == 1: A[k + m lda] (A is K x M).  Eight elements per thread and stage; the
  // addresses of stage 0 are formed once and advance by 16 columns (TA == 0) / 16 rows (TA == 1) per stage; rows / columns outside the matrix
  // are clamped to a valid address and their values replaced by zero.
  double ra[8], rb[8];
  const double *pa[8], *pb[8];
  bool va[8], vb[8];
  int ka[8], kb[8];
#pragma unroll
  for (int i = 0; i < 8; i++) {
    long m, k, n;
    if (TA == 0) { m = m0 + (t & 127); k = (t >> 7) * 8 + i; } else { k = t & 15; m = m0 + (t >> 4) + 16 * i; }
    va[i] = m < M; ka[i] = (int)k;
    if (!va[i]) m = 0;
    pa[i] = TA == 0 ? A + m + k * lda : A + k + m * lda;
    if (TB == 0) { k = t & 15; n = n0 + (t >> 4) + 16 * i; } else { n = n0 + (t & 127); k = (t >> 7) * 8 + i; }
    vb[i] = n < N; kb[i] = (int)k;
    if (!vb[i]) n = 0;
    pb[i] = TB == 0 ? B + k + n * ldb : B + n + k * ldb;
  }
  const long stepA = TA == 0 ? kDgBK * lda : kDgBK, stepB = TB == 0 ? kDgBK : kDgBK * ldb;
  auto load_stage = [&](long k0) {
    if (k0 + kDgBK <= K) {
#pragma unroll
      for (int i = 0; i < 8; i++) { ra[i] = va[i] ? *pa[i] : 0.0; rb[i] = vb[i] ? *pb[i] : 0.0; }
    } else {
#pragma unroll
      for (int i = 0; i < 8; i++) { ra[i] = (va[i] && k0 + ka[i] < K) ? *pa[i] : 0.0; rb[i] = (vb[i] && k0 + kb[i] < K) ? *pb[i] : 0.0; }
    }
#pragma unroll
    for (int i = 0; i < 8; i++) { pa[i] += stepA; pb[i] += stepB; }
  };
  auto store_stage = [&]() {
#pragma unroll
    for (int i = 0; i < 8; i++) {
      if (TA == 0) As[(t >> 7) * 8 + i][t & 127] = ra[i]; else As[t & 15][(t >> 4) + 16 * i] = ra[i];
      if (TB == 0) Bs[t & 15][(t >> 4) + 16 * i] = rb[i]; else Bs[(t >> 7) * 8 + i][t & 127] = rb[i];
    }
  };
  double4_t acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; a++)
#pragma unroll
    for (int b = 0; b < 4; b++) acc[a][b] = double4_t{0.0, 0.0, 0.0, 0.0};
  const long steps = (K + kDgBK - 1) / kDgBK;
  // MFMA tiles of this wave that lie inside the matrix (wave-uniform): small operands (the 64-row steps of the triangular solves, a few
  // right-hand sides) do not pay for a full 128 x 128 tile; lower_only: a wave tile strictly above the diagonal is skipped as well
  int na = (int)((M - m0 - wm * 64 + 15) / 16), nbt = (int)((N - n0 - wn * 64 + 15) / 16);
  na = na < 0 ? 0 : (na > 4 ? 4 : na); nbt = nbt < 0 ? 0 : (nbt > 4 ? 4 : nbt);
  if (lower_only && n0 + wn * 64 > m0 + wm * 64 + 63) na = 0;
  load_stage(0);
  store_stage();
  __syncthreads();
  for (long s = 0; s < steps; s++) {
    if (s + 1 < steps) load_stage((s + 1) * kDgBK);   // global -> registers, in flight under the MFMAs of this stage
#pragma unroll
    for (int kk = 0; kk < 4; kk++) {
      double fa[4], fb[4];
      const int kr = kk * 4 + (lane >> 4);
#pragma unroll
      for (int a = 0; a < 4; a++) fa[a] = As[kr][wm * 64 + a * 16 + (lane & 15)];
#pragma unroll
      for (int b = 0; b < 4; b++) fb[b] = Bs[kr][wn * 64 + b * 16 + (lane & 15)];
#pragma unroll
      for (int a = 0; a < 4; a++)
#pragma unroll
        for (int b = 0; b < 4; b++)
          if (a < na && b < nbt) acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[a], fb[b], acc[a][b], 0, 0, 0);
    }
    __syncthreads();
    if (s + 1 < steps) store_stage();
    __syncthreads();
  }
  // D layout of v_mfma_f64_16x16x4_f64: column = lane & 15, row = (lane >> 4) + 4 r
#pragma unroll
  for (int a = 0; a < 4; a++)
#pragma unroll
    for (int b = 0; b < 4; b++) {
      if (!(a < na && b < nbt)) continue;
      const long n = n0 + wn * 64 + b * 16 + (lane & 15);
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const long m = m0 + wm * 64 + a * 16 + (lane >> 4) + 4 * r;
        if (m < M && n < N) {
          double *c = C + m + n * ldc;
          *c = beta == 0.0 ? alpha * acc[a][b][r] : fma(alpha, acc[a][b][r], beta * *c);
        }
      }
    }
}

int launch_dgemm(bool ta, bool tb, long M, long N, long K, double alpha, const double *A, long lda, const double *B, long ldb, double beta, double *C, long ldc,
                 bool lower_only, hipStream_t s) {
  if (M <= 0 || N <= 0) return 0;
  const dim3 grid((unsigned)((M + kDgBM - 1) / kDgBM), (unsigned)((N + kDgBN - 1) / kDgBN)), block(256);
  const int lo = lower_only ? 1 : 0;
  if (!ta && !tb) hipLaunchKernelGGL((k_dgemm<0, 0>), grid, block, 0, s, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, lo);
  else if (!ta && tb) hipLaunchKernelGGL((k_dgemm<0, 1>), grid, block, 0, s, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, lo);
  else if (ta && !tb) hipLaunchKernelGGL((k_dgemm<1, 0>), grid, block, 0, s, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, lo);
  else hipLaunchKernelGGL((k_dgemm<1, 1>), grid, block, 0, s, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, lo);
  MXA_HIP(hipGetLastError());
  return 0;
}

// Cholesky of one nb x nb (nb <= 64) diagonal block (lower triangle, column-major, ld) AND the inverse of its factor, one workgroup, in LDS.
// Factor: right-looking; per column one barrier -- every thread updates its part of the trailing lower triangle with the UNSCALED column
// (a_ik -= a_ij a_kj / d), then one wave writes the scaled column behind the barrier (nobody reads column j again).  Thread (row = tid & 63,
// tid >> 6) walks k = j + 1 + (tid >> 6), step 4: no integer divisions (round 2's idx % t, idx / t took most of its 90 us per block).
// Inverse: four 16 x 16 diagonal blocks by forward substitution (16 threads each), then two merge levels
// X21 = -X22 (L21 X11) as small products over all 256 threads.  inv_out: 64 x 64, column-major, zero outside the nb x nb lower triangle.
// *info = (1-based global index of the first non-positive pivot) if the block is not positive definite, untouched otherwise.
__global__ void __launch_bounds__(256) k_potrf_inv_block(double *__restrict__ A, long ld, int nb, long offset, int *__restrict__ info, double *__restrict__ inv_out) {
  __shared__ double a[64][65], x[64][65], tmp[64][65];
  const int tid = threadIdx.x, ti = tid & 63, tk = tid >> 6;
  for (int c = tk; c < 64; c += 4) {
    a[ti][c] = (ti < nb && c < nb) ? (ti >= c ? A[ti + (long)c * ld] : 0.0) : (ti == c ? 1.0 : 0.0);   // identity padding keeps the inverse finite
    x[ti][c] = 0.0;
  }
  __syncthreads();
  for (int j = 0; j < nb; j++) {
    const double d = a[j][j];
    if (!(d > 0.0)) { if (tid == 0) atomicCAS(info, 0, (int)(offset + j + 1)); return; }   // uniform: every thread reads the same d
    const double cij = a[ti][j], rd = 1.0 / d;
    if (ti > j && ti < nb)
      for (int k = j + 1 + tk; k <= ti; k += 4) a[ti][k] -= cij * a[k][j] * rd;
    __syncthreads();
    if (tk == 0 && ti >= j && ti < nb) { const double l = sqrt(d); a[ti][j] = ti == j ? l : cij / l; }
  }
  __syncthreads();
  for (int c = tk; c < nb; c += 4) if (ti < nb && ti >= c) A[ti + (long)c * ld] = a[ti][c];
  // inverse of the 16 x 16 diagonal blocks: thread (block q, column jj) does column 16 q + jj by forward substitution
  if (tid < 64) {
    const int q = tid >> 4, j = tid, r1 = 16 * q + 16;
    x[j][j] = 1.0 / a[j][j];
    for (int i = j + 1; i < r1; i++) {
      double sum = 0.0;
      for (int k = j; k < i; k++) sum = fma(a[i][k], x[k][j], sum);
      x[i][j] = -sum / a[i][i];
    }
  }
  __syncthreads();
  for (int size = 16; size < 64; size *= 2) {
    const int lg = size == 16 ? 4 : 5, outs = 32 * size;            // (64 / (2 size)) merges x size^2 outputs
    // tmp = L21 X11 (X11 lower triangular)
    for (int o = tid; o < outs; o += 256) {
      const int mrg = o >> (2 * lg), rem = o & (size * size - 1), ii = rem & (size - 1), jj = rem >> lg, r0 = mrg * 2 * size;
      double sum = 0.0;
      for (int k = jj; k < size; k++) sum = fma(a[r0 + size + ii][r0 + k], x[r0 + k][r0 + jj], sum);
      tmp[r0 + size + ii][r0 + jj] = sum;
    }
    __syncthreads();
    // X21 = -X22 tmp (X22 lower triangular)
    for (int o = tid; o < outs; o += 256) {
      const int mrg = o >> (2 * lg), rem = o & (size * size - 1), ii = rem & (size - 1), jj = rem >> lg, r0 = mrg * 2 * size;
      double sum = 0.0;
      for (int k = 0; k <= ii; k++) sum = fma(x[r0 + size + ii][r0 + size + k], tmp[r0 + size + k][r0 + jj], sum);
      x[r0 + size + ii][r0 + jj] = -sum;
    }
    __syncthreads();
  }
  for (int c = tk; c < 64; c += 4) inv_out[ti + c * 64] = (ti < nb && c < nb) ? x[ti][c] : 0.0;
}

int launch_potrf_inv_block(double *A, long ld, int nb, long offset, int *info, double *inv_out, hipStream_t s) {
  hipLaunchKernelGGL(k_potrf_inv_block, dim3(1), dim3(256), 0, s, A, ld, nb, offset, info, inv_out);
  MXA_HIP(hipGetLastError());
  return 0;
}

}  // namespace mxa
