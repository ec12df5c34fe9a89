/* oracle.c -- CPU restatement of the reference's compressed-genotype GEMM path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library, and only as
 * the checker / the CPU baseline.  The product (miraculix_amd/csrc) never links or calls it.
 *
 * Parity status: PINNED.  This restatement is checked (tests/test_oracle.py::test_port_vs_live_reference_library, run in the
 * build container where /root/reference exists) against the reference's own CPU library compiled
 * from its sources by oracle/Makefile.ref, and against the committed fixtures tests/golden/ that
 * were emitted by that library (script: tests/golden/make_golden.py).
 *
 * Every function cites the reference file:line it follows (paths relative to /root/reference).
 *
 * Plain C (gcc -O3 -fopenmp).  All matrices B, C are column-major fp64 like the reference ABI
 * (src/miraculix/5codesAPI.c:98-110).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------------------------------ */
/* PLINK .bed 2-bit code -> allele count.  00->0, 01 (missing)->0, 10->1, 11->2, i.e.
 * value = max(code-1, 0): src/cuda/dgemm_compressed_cuda.h:259-266, src/miraculix/MXinfo.h:143.
 * Genotype i of a row sits in bits 2*(i%4) of byte i/4 (src/bindings/Julia/read_plink.jl:152). */
static inline int plink_value(int code) { return code > 1 ? code - 1 : 0; }
static inline int get_code(const uint8_t *row, long i) { return (row[i >> 2] >> (2 * (i & 3))) & 3; }

/* Z[i][s] as a dense int8 matrix from the SNP-major payload (snps rows of ceil(indiv/4) bytes) */
void oracle_decode_snpmajor(const uint8_t *plink, long snps, long indiv, int8_t *Z /* indiv x snps row-major */) {
  long bps = (indiv + 3) / 4;
  for (long s = 0; s < snps; s++)
    for (long i = 0; i < indiv; i++) Z[i * snps + s] = (int8_t)plink_value(get_code(plink + s * bps, i));
}

/* 2-bit transpose: `rows` rows of ceil(cols/4) bytes  ->  `cols` rows of ceil(rows/4) bytes, padding bits zero.
 * Follows src/bindings/Julia/compressed_operations.jl:45-66 (transpose_genotype_matrix) and
 * src/bindings/Fortran/modplink_miraculix.f90:177-218 (transpose_integermatrix). */
void oracle_transpose_2bit(const uint8_t *in, long rows, long cols, uint8_t *out) {
  long bin = (cols + 3) / 4, bout = (rows + 3) / 4;
  memset(out, 0, (size_t)cols * bout);
  for (long r = 0; r < rows; r++)
    for (long c = 0; c < cols; c++) {
      int code = get_code(in + r * bin, c);
      out[c * bout + (r >> 2)] |= (uint8_t)(code << (2 * (r & 3)));
    }
}

/* allele frequency by popcount: f_s = (#set bits of SNP s, missing 01 counted as 0) / (2*indiv).
 * src/bindings/Julia/read_plink.jl:199-203 computes popcount(bytes)/(2n) on missing-free data; for
 * codes {00,10,11} popcount == allele count, which is what is restated here. */
void oracle_allele_freq(const uint8_t *plink, long snps, long indiv, double *f) {
  long bps = (indiv + 3) / 4;
  for (long s = 0; s < snps; s++) {
    long cnt = 0;
    for (long i = 0; i < indiv; i++) cnt += plink_value(get_code(plink + s * bps, i));
    f[s] = (double)cnt / (2.0 * (double)indiv);
  }
}

/* ------------------------------------------------------------------------------------------ */
/* (a) Analytic oracle: C = op(Z - 2*1*f^T) * B with long-double accumulation.
 * This is the dense oracle of the reference's own tests: tests/dgemm_compressed/test.jl:97-104
 * ((G .- 2f) * B) and test_5codesapi.f90:197-210.  trans=0: C(indiv x n) = Zc * B(snps x n);
 * trans=1: C(snps x n) = Zc^T * B(indiv x n)   (docs/genotype_matrix_multiplication.md:30-40).
 * Missing (01) decodes to 0 and is then centred like any other entry. */
void oracle_dgemm_dense(int trans, const uint8_t *plink, long snps, long indiv, const double *f, int centered,
                        long n, const double *B, long ldb, double *C, long ldc) {
  long bps = (indiv + 3) / 4;
  long m = trans ? snps : indiv;
#pragma omp parallel for schedule(static)
  for (long j = 0; j < n; j++) {
    const double *b = B + j * ldb;
    double *c = C + j * ldc;
    if (trans) {
      for (long s = 0; s < snps; s++) {
        long double acc = 0.0L;
        long double fs = centered ? 2.0L * (long double)f[s] : 0.0L;
        const uint8_t *row = plink + s * bps;
        for (long i = 0; i < indiv; i++) acc += ((long double)plink_value(get_code(row, i)) - fs) * (long double)b[i];
        c[s] = (double)acc;
      }
    } else {
      long double *acc = (long double *)calloc((size_t)indiv, sizeof(long double));
      for (long s = 0; s < snps; s++) {
        long double fs = centered ? 2.0L * (long double)f[s] : 0.0L;
        long double bs = (long double)b[s];
        const uint8_t *row = plink + s * bps;
        for (long i = 0; i < indiv; i++) acc[i] += ((long double)plink_value(get_code(row, i)) - fs) * bs;
      }
      for (long i = 0; i < indiv; i++) c[i] = (double)acc[i];
      free(acc);
    }
    for (long r = m; r < ldc; r++) c[r] = 0.0; /* CPU path zero-fills the ld padding: 5codesIntern.h:67 */
  }
}

/* (b) Restatement of the reference GPU arithmetic (the only arithmetic definition of src/cuda):
 * per packed byte 4 fp64 FMAs in ascending field order, bytes in ascending order
 * (dgemm_compressed_cuda.h:253-268), then the centring epilogue in plain fp64:
 * w_j = -2 * sum_k B[k,j]*x_k (cublasDgemv, dgemm_compressed_cuda.cu:426-437), C[:,j] += w_j * y
 * (cublasDaxpy, :449-459) with (x,y) = (f,1) for 'N' and (1,f) for 'T'.  K order inside CUTLASS'
 * threadblock tiling is ascending too, so this is the fp64 result the CUDA path produces up to
 * cuBLAS' internal dgemv summation order. */
void oracle_dgemm_gpuorder(int trans, const uint8_t *plink, const uint8_t *plink_t, long snps, long indiv,
                           const double *f, int centered, long n, const double *B, long ldb, double *C, long ldc) {
  const uint8_t *A = trans ? plink : plink_t; /* dgemm_compressed_cuda.cu:270 */
  long m = trans ? snps : indiv, k = trans ? indiv : snps, k1 = (k + 3) / 4;
#pragma omp parallel for schedule(static)
  for (long j = 0; j < n; j++) {
    const double *b = B + j * ldb;
    double *c = C + j * ldc;
    for (long r = 0; r < m; r++) {
      double d = 0.0;
      const uint8_t *row = A + r * k1;
      for (long kb = 0; kb < k1; kb++)
        for (int q = 0; q < 4; q++) {
          long kk = 4 * kb + q;
          double bv = kk < k ? b[kk] : 0.0; /* zero-padded B rows: dgemm_compressed_cuda.cu:300-317 */
          d = fma((double)plink_value((row[kb] >> (2 * q)) & 3), bv, d);
        }
      c[r] = d;
    }
    if (centered) {
      double w = 0.0;
      for (long kk = 0; kk < k; kk++) w += b[kk] * (trans ? 1.0 : f[kk]);
      w *= -2.0;
      for (long r = 0; r < m; r++) c[r] += w * (trans ? f[r] : 1.0);
    }
  }
}

/* ------------------------------------------------------------------------------------------ */
/* (c) "5codes" CPU engine, restated from scratch.
 *   - base-3 packing, 5 genotypes per byte, byte = sum_k g_k 3^k, k=0 least significant
 *     (5codesUint.cc:55-101 initiate_table5I / PLINK2FIVE);
 *   - two code matrices so that the kernel is always "genoVector"-shaped, ans[row] = sum_col G[row,col] v[col]:
 *     layout [column-group][row] one byte each (5codesChar.cc:213-340, TemplateUint.h:164-176);
 *   - per group of 5 RHS entries a 243-entry table T[c] = sum_k digit_k(c) * v_k built in long double and
 *     stored as double (5codesIntern.h:130-184 gV5_CreateHash);
 *   - main loop t[row] += (T0[c0]+T1[c1]) + (T2[c2]+T3[c3]) four column groups at a time, the column
 *     groups cut into `blocks` slices that accumulate into separate partial vectors (5codesIntern.h:57-100, 214-266);
 *   - slices summed 4-way as a tree (5codesIntern.h:321-342 gV5_SumUp), scatter to Ans (:371-386);
 *   - RowMeans centring as rank-1 long-double corrections (Vector.matrix.D.cc:101-114,145-175),
 *     external allele frequencies used verbatim (haplogeno.cc:1591-1593).
 */
typedef struct {
  long snps, indiv;
  uint8_t *code_T; /* 'T' (genoVector): rows = snps,  column groups over individuals: [ceil(indiv/5) padded][snps]  */
  uint8_t *code_N; /* 'N' (vectorGeno): rows = indiv, column groups over SNPs:        [ceil(snps/5) padded][indiv] */
  long groups_T, groups_N; /* padded to a multiple of 4 */
  long double *f;
  int cores;
} oracle5_t;

static long div_geq(long a, long b) { return (a + b - 1) / b; }

void *oracle5_create(const uint8_t *plink, long snps, long indiv, const double *f, int cores) {
  oracle5_t *h = (oracle5_t *)calloc(1, sizeof(oracle5_t));
  long bps = (indiv + 3) / 4;
  h->snps = snps; h->indiv = indiv; h->cores = cores > 0 ? cores : 1;
  h->groups_T = div_geq(div_geq(indiv, 5), 4) * 4;
  h->groups_N = div_geq(div_geq(snps, 5), 4) * 4;
  h->code_T = (uint8_t *)calloc((size_t)h->groups_T * snps, 1);
  h->code_N = (uint8_t *)calloc((size_t)h->groups_N * indiv, 1);
  static const int pow3[5] = {1, 3, 9, 27, 81};
#pragma omp parallel for schedule(static) num_threads(h->cores)
  for (long s = 0; s < snps; s++) {
    const uint8_t *row = plink + s * bps;
    for (long i = 0; i < indiv; i++) {
      int g = plink_value(get_code(row, i));
      if (g) h->code_T[(i / 5) * snps + s] += (uint8_t)(g * pow3[i % 5]);
    }
  }
#pragma omp parallel for schedule(static) num_threads(h->cores)
  for (long i = 0; i < indiv; i++) {
    for (long s = 0; s < snps; s++) {
      int g = plink_value(get_code(plink + s * bps, i));
      if (g) h->code_N[(s / 5) * indiv + i] += (uint8_t)(g * pow3[s % 5]);
    }
  }
  h->f = (long double *)calloc((size_t)snps, sizeof(long double));
  if (f) for (long s = 0; s < snps; s++) h->f[s] = (long double)f[s];
  return h;
}

void oracle5_free(void *hh) {
  oracle5_t *h = (oracle5_t *)hh;
  if (!h) return;
  free(h->code_T); free(h->code_N); free(h->f); free(h);
}

/* genoVector kernel for up to VATONCE = 4 right-hand side columns per pass over the code matrix (the reference interleaves VatOnce = 4 columns in
 * one AVX2 register, 5codesIntern.h:130-266; each column's arithmetic is independent and identical to the one-column scalar form, so the
 * results do not depend on how the columns are grouped -- checked bit for bit against the reference library in tests/test_oracle.py). */
/* Columns per pass: 1.  Four per pass (the reference's VatOnce) is 1.5x faster on the build container's Xeon but no faster on the GPU hosts' EPYC
 * 9575F, where the baseline is timed (16 cores, 100k x 50k x 32 sample: N 0.437 s / T 0.476 s against 0.423 / 0.437 with one column per pass). */
#ifndef GV5_VATONCE
#define GV5_VATONCE 1
#endif
/* scratch of a product: the 243-entry tables and the partial vectors, allocated once per oracle5_dgemm call and reused by its passes (the
 * reference allocates them once per call for all columns, 5codesIntern.h:96-128; a fresh calloc per pass costs the page faults of 64 MB each time) */
typedef struct { double *F; size_t f_len; double *Tmp; size_t tmp_len; } gv5_scratch;
static void gv5_kernel(const uint8_t *code, long rows, long cols, long groups_padded, const double *const *v, int nv, int cores,
                       double *const *ans, gv5_scratch *sc) {
  const long colsCpB = div_geq(cols, 5);
  const long colBlocks = div_geq(colsCpB, 4);
  long blockSliceLen = div_geq(colBlocks, (long)cores * 5); /* coreFactor 5: 5codesIntern.h:54 */
  if (blockSliceLen > 100) blockSliceLen = 100;             /* SLICELEN */
  if (blockSliceLen < 1) blockSliceLen = 1;
  const long blocks = div_geq(colBlocks, blockSliceLen);
  const long sliceLen = blockSliceLen * 4;
  const long rest = (colBlocks - blockSliceLen * (blocks - 1)) * 4;
  (void)groups_padded;
  /* hash tables: [group][column q][243] so that the nv tables of a group are neighbours */
  const size_t fstride = (size_t)nv * 243;
  const size_t f_len = (size_t)colBlocks * 4 * fstride;
  if (sc->f_len < f_len) { free(sc->F); sc->F = (double *)malloc(f_len * sizeof(double)); sc->f_len = f_len; }
  double *F = sc->F;
#pragma omp parallel for schedule(static) num_threads(cores)
  for (long i = colsCpB; i < colBlocks * 4; i++) memset(F + (size_t)i * fstride, 0, fstride * sizeof(double));   /* tables of the padding groups */
#pragma omp parallel for schedule(static) num_threads(cores)
  for (long i = 0; i < colsCpB; i++) {
    for (int q = 0; q < nv; q++) {
      long double x[5] = {0, 0, 0, 0, 0};
      long have = (i < colsCpB - 1) ? 5 : cols - (colsCpB - 1) * 5;
      for (long k = 0; k < have; k++) x[k] = (long double)v[q][i * 5 + k];
      double *hash = F + (size_t)i * fstride + (size_t)q * 243;
      for (int i4 = 0; i4 < 3; i4++) {
        long double f4 = (long double)i4 * x[4];
        for (int i3 = 0; i3 < 3; i3++) {
          long double f3 = f4 + (long double)i3 * x[3];
          for (int i2 = 0; i2 < 3; i2++) {
            long double f2 = f3 + (long double)i2 * x[2];
            for (int i1 = 0; i1 < 3; i1++) {
              long double f1 = f2 + (long double)i1 * x[1];
              int V0 = 3 * (3 * (3 * (3 * i4 + i3) + i2) + i1);
              long double f0 = f1;
              hash[V0] = (double)f0;
              f0 += x[0];
              hash[V0 + 1] = (double)f0;
              hash[V0 + 2] = (double)(f0 + x[0]);
            }
          }
        }
      }
    }
  }
  /* main loop: one partial vector per slice and column (+1 zero slab per column used by the tree) */
  const size_t tstride = (size_t)(blocks + 1) * rows;
  if (sc->tmp_len < (size_t)nv * tstride) { free(sc->Tmp); sc->Tmp = (double *)malloc((size_t)nv * tstride * sizeof(double)); sc->tmp_len = (size_t)nv * tstride; }
  double *Tmp = sc->Tmp;
#pragma omp parallel for schedule(static) num_threads(cores)
  for (long Cb = 0; Cb <= blocks; Cb++)     /* every thread clears the slabs it is about to accumulate into (and the zero slab the tree reads) */
    for (int q = 0; q < nv; q++) memset(Tmp + (size_t)q * tstride + rows * Cb, 0, (size_t)rows * sizeof(double));
  const long RoughRowChunk = 35000;
  long rowBlocks = rows / RoughRowChunk; if (rowBlocks < 1) rowBlocks = 1;
  const long RowChunk = div_geq(rows, rowBlocks);
  for (long bStart = 0; bStart < rows; bStart += RowChunk) {
    long bEnd = bStart + RowChunk < rows ? bStart + RowChunk : rows;
#pragma omp parallel for schedule(static) num_threads(cores)
    for (long Cb = 0; Cb < blocks; Cb++) {
      const double *ff = F + (size_t)sliceLen * Cb * fstride;
      const uint8_t *c = code + rows * sliceLen * Cb;
      long nrCols = Cb == blocks - 1 ? rest : sliceLen;
      for (long i = 0; i < nrCols; i += 4) {
        const uint8_t *p0 = c + (i + 0) * rows, *p1 = c + (i + 1) * rows, *p2 = c + (i + 2) * rows, *p3 = c + (i + 3) * rows;
        const double *g0 = ff + (size_t)(i + 0) * fstride, *g1 = ff + (size_t)(i + 1) * fstride, *g2 = ff + (size_t)(i + 2) * fstride, *g3 = ff + (size_t)(i + 3) * fstride;
        if (GV5_VATONCE == 4 && nv == 4) {   /* the code bytes are loaded once for the four columns */
          double *t0 = Tmp + rows * Cb, *t1 = t0 + tstride, *t2 = t1 + tstride, *t3 = t2 + tstride;
          for (long b = bStart; b < bEnd; b++) {
            const int c0 = p0[b], c1 = p1[b], c2 = p2[b], c3 = p3[b];
            t0[b] += (g0[c0] + g1[c1]) + (g2[c2] + g3[c3]);
            t1[b] += (g0[243 + c0] + g1[243 + c1]) + (g2[243 + c2] + g3[243 + c3]);
            t2[b] += (g0[486 + c0] + g1[486 + c1]) + (g2[486 + c2] + g3[486 + c3]);
            t3[b] += (g0[729 + c0] + g1[729 + c1]) + (g2[729 + c2] + g3[729 + c3]);
          }
        } else {
          for (int q = 0; q < nv; q++) {
            double *t = Tmp + (size_t)q * tstride + rows * Cb;
            const double *f0 = g0 + q * 243, *f1 = g1 + q * 243, *f2 = g2 + q * 243, *f3 = g3 + q * 243;
            for (long b = bStart; b < bEnd; b++) t[b] += (f0[p0[b]] + f1[p1[b]]) + (f2[p2[b]] + f3[p3[b]]);
          }
        }
      }
    }
  }
  /* 4-way tree sum of the slices, per column (5codesIntern.h:321-341).  The reference runs this loop in parallel over its groups of columns
   * (:408-411); here every pass handles one group, so the rows are dealt to the threads instead -- each thread walks the whole tree for its
   * own rows: the same additions in the same order for every element. */
  const long blocksXrows = blocks * rows;
  const long chunk = 4096;
#pragma omp parallel for schedule(static) num_threads(cores)
  for (long r0 = 0; r0 < rows; r0 += chunk) {
    const long r1 = r0 + chunk < rows ? r0 + chunk : rows;
    for (int q = 0; q < nv; q++) {
      double *T = Tmp + (size_t)q * tstride;
      long level = rows, tmpCols = blocks;
      while (tmpCols > 1) {
        long tmpC4 = div_geq(tmpCols - 1, 4);
        for (long k = 0; k < tmpC4; k++) {
          long kS = k * 4;
          long o0 = (kS + 0) * level, o1 = (kS + 1) * level, o2 = (kS + 2) * level, o3 = (kS + 3) * level;
          double *t0 = T + (o0 < blocksXrows ? o0 : blocksXrows), *t1 = T + (o1 < blocksXrows ? o1 : blocksXrows);
          double *t2 = T + (o2 < blocksXrows ? o2 : blocksXrows), *t3 = T + (o3 < blocksXrows ? o3 : blocksXrows);
          for (long j = r0; j < r1; j++) t0[j] = (t0[j] + t1[j]) + (t2[j] + t3[j]);
        }
        level *= 4;
        tmpCols = div_geq(tmpCols, 4);
      }
      for (long bb = r0; bb < r1; bb++) ans[q][bb] = T[bb];
    }
  }
}

void oracle5_dgemm(void *hh, int trans, int centered, long n, const double *B, long ldb, double *C, long ldc) {
  oracle5_t *h = (oracle5_t *)hh;
  const long rows = trans ? h->snps : h->indiv, cols = trans ? h->indiv : h->snps;
  const uint8_t *code = trans ? h->code_T : h->code_N;
  const long groups = trans ? h->groups_T : h->groups_N;
  memset(C, 0, (size_t)ldc * n * sizeof(double)); /* 5codesIntern.h:67 */
  gv5_scratch sc = {0, 0, 0, 0};
  for (long j0 = 0; j0 < n; j0 += GV5_VATONCE) {
    const int nv = (int)(n - j0 < GV5_VATONCE ? n - j0 : GV5_VATONCE);
    const double *vv[GV5_VATONCE];
    double *aa[GV5_VATONCE];
    for (int q = 0; q < nv; q++) { vv[q] = B + (j0 + q) * ldb; aa[q] = C + (j0 + q) * ldc; }
    gv5_kernel(code, rows, cols, groups, vv, nv, h->cores, aa, &sc);
  }
  free(sc.F); free(sc.Tmp);
  for (long j = 0; j < n; j++) {
    const double *v = B + j * ldb;
    double *a = C + j * ldc;
    if (centered) { /* Vector.matrix.D.cc:101-114 and :145-175 with meanV=meanSxI=false, RowMeans */
      if (trans) {
        long double s = 0.0L;
        for (long i = 0; i < cols; i++) s += (long double)v[i];
        s *= -2.0L;
        for (long r = 0; r < rows; r++) a[r] = (double)((long double)a[r] + s * h->f[r]);
      } else {
        long double fv = 0.0L;
        for (long s = 0; s < cols; s++) fv += h->f[s] * (long double)v[s];
        long double ones = -1.0L * fv * 2.0L;
        for (long r = 0; r < rows; r++) a[r] = (double)((long double)a[r] + ones);
      }
    }
  }
}

/* ------------------------------------------------------------------------------------------ */
/* (d) Integer crossproduct oracle: ans = X * X^T, X = `rows` x `k` 2-bit values, row = ceil(k/4) bytes.
 * Arithmetic: exact sum over k of (2-bit value)*(2-bit value) (snp_multiply_cuda.h:121-199: two u4 MMAs on the
 * low/high 2-bit field of each nibble == sum of products of the 2-bit fields).  is_plink_format applies the
 * reference's byte LUT first (snp_multiply_cuda.h:202-210): 00->0, 10->1, 11->2 and ANY byte containing a
 * missing pair (01) becomes 0xFF, i.e. all four genotypes of that byte read as 3.
 * Argument naming follows the reference's positional meaning: `k` = packed dimension,
 * `rows` = output dimension.  Padding bits of the last byte take part exactly as stored (the reference
 * multiplies whole bytes); valid bed files have them zero. */
static uint8_t plink_lut_byte(uint8_t v) {
  uint8_t out = 0;
  for (int q = 0; q < 4; q++) {
    int code = (v >> (2 * q)) & 3;
    if (code == 1) return 0xFF;
    out |= (uint8_t)((code ? code - 1 : 0) << (2 * q));
  }
  return out;
}

void oracle_crossprod_i32(const uint8_t *X, long k, long rows, int is_plink_format, int32_t *ans /* rows x rows */) {
  long bpr = (k + 3) / 4;
  uint8_t *Y = (uint8_t *)malloc((size_t)rows * bpr);
  for (long i = 0; i < rows * bpr; i++) Y[i] = is_plink_format ? plink_lut_byte(X[i]) : X[i];
  /* byte x byte product table: sum over the 4 fields */
  static int16_t tab[256][256];
  static int tab_ready = 0;
  if (!tab_ready) {
    for (int a = 0; a < 256; a++)
      for (int b = 0; b < 256; b++) {
        int s = 0;
        for (int q = 0; q < 4; q++) s += ((a >> (2 * q)) & 3) * ((b >> (2 * q)) & 3);
        tab[a][b] = (int16_t)s;
      }
    tab_ready = 1;
  }
#pragma omp parallel for schedule(dynamic, 8)
  for (long i = 0; i < rows; i++) {
    for (long j = i; j < rows; j++) {
      const uint8_t *a = Y + i * bpr, *b = Y + j * bpr;
      int64_t s = 0;
      for (long t = 0; t < bpr; t++) s += tab[a[t]][b[t]];
      ans[i * rows + j] = (int32_t)s;
      ans[j * rows + i] = (int32_t)s;
    }
  }
  free(Y);
}

/* the ABI returns doubles, column-major rows x rows, full symmetric (snp_multiply_cuda.cu:330-340) */
void oracle_crossprod_f64(const uint8_t *X, long k, long rows, int is_plink_format, double *ans) {
  int32_t *t = (int32_t *)malloc((size_t)rows * rows * sizeof(int32_t));
  oracle_crossprod_i32(X, k, rows, is_plink_format, t);
  for (long i = 0; i < rows * rows; i++) ans[i] = (double)t[i];
  free(t);
}

int oracle_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

/* ------------------------------------------------------------------------------------------ */
/* sparse_times_plink restated (reference: src/miraculix/5codesAPI.c:135-157 -> sparseTGenoPlinkApi 5codesChar.cc:472-491 ->
 * sparseTGeno haplogeno.cc:1683-1711 -> sparseTGenoPlink plinkUint.cc:352-470).  The sparse column index selects a ROW of the
 * packed matrix P (rows x ceil(entries/4) bytes), the result runs over that row's 2-bit entries:
 *   C[j + e*ldc] = sum_{t in row j of S} val[t] * z(P[col[t]], e),   z = 00->0, 01->0, 10->1, 11->2,
 * zero-based CSR, C zero-filled over ldc x entries first (haplogeno.cc:1696).  The reference adds the stored entries in groups of
 * up to 8 (its MULTI switch); here they are accumulated in long double, so agreement is to rounding, not bitwise.
 * Pinned against the reference library's own output in tests/golden/sparse_golden.npz (tests/golden/make_golden_sparse.py). */
void oracle_sparse_times_plink(const uint8_t *P, long rows, long entries, int nIdx, const int *rowIdx, const int *colIdx,
                               const double *val, double *C, long ldc) {
  long bps = (entries + 3) / 4;
  (void)rows;
  for (long i = 0; i < ldc * entries; i++) C[i] = 0.0;
#pragma omp parallel for schedule(static)
  for (long e = 0; e < entries; e++) {
    for (int j = 0; j < nIdx; j++) {
      long double acc = 0.0L;
      for (int t = rowIdx[j]; t < rowIdx[j + 1]; t++)
        acc += (long double)val[t] * (long double)plink_value(get_code(P + (long)colIdx[t] * bps, e));
      C[j + e * ldc] = (double)acc;
    }
  }
}
