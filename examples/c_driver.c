/* c_driver.c -- the reference's Fortran integration test (tests/dgemm_compressed/test_5codesapi.f90:170-210 and its
 * '_t' twin) restated as a plain C program against the C ABI: synthetic PLINK data, ncol = 10 ('n') / 15 ('t'),
 * B = -(10 i + j) resp. -(1000 i + j), centred by f, result compared with a dense matmul on the decoded genotypes,
 * failure if any |difference| > 1e-4 (the reference's own acceptance threshold; observed ~1e-10).
 * Then one call each of the other reference entries: snp_multiply_gpu, sparse_times_plink, potrs_solve_gpu, sparse2gpu /
 * dcsrtrsv_solve_gpu / free_sparse_gpu, each checked against a loop in this file.
 * Shows that compiled callers only need to link libmiraculix_amd.so:
 *   gcc -O2 -Iinclude examples/c_driver.c -o c_driver -Lmiraculix_amd/lib -lmiraculix_amd -Wl,-rpath,$PWD/miraculix_amd/lib -lm
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "miraculix_amd.h"

static unsigned long long rng_state = 88172645463325252ULL;
static unsigned rnd(void) { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return (unsigned)(rng_state >> 32); }

int main(int argc, char **argv) {
  const int snps = argc > 1 ? atoi(argv[1]) : 1000, indiv = argc > 2 ? atoi(argv[2]) : 500;   /* BASELINE config 1 */
  const int bps = (indiv + 3) / 4, bpi = (snps + 3) / 4;
  unsigned char *Z = malloc((size_t)snps * indiv);
  char *plink = calloc((size_t)snps * bps, 1), *plink_t = calloc((size_t)indiv * bpi, 1);
  double *f = calloc(snps, sizeof(double));
  for (int s = 0; s < snps; s++)
    for (int i = 0; i < indiv; i++) {
      int z = (rnd() % 3 + rnd() % 3) / 2;              /* 0,1,2 */
      Z[(size_t)s * indiv + i] = (unsigned char)z;
      int code = z == 0 ? 0 : z + 1;                    /* 0->00, 1->10, 2->11 */
      plink[(size_t)s * bps + i / 4] |= (char)(code << (2 * (i % 4)));
      plink_t[(size_t)i * bpi + s / 4] |= (char)(code << (2 * (s % 4)));
      f[s] += z;
    }
  for (int s = 0; s < snps; s++) f[s] /= 2.0 * indiv;

  setOptions_compressed(/*use_gpu*/1, 0, 0, 0, /*ignore_missings*/1, /*do_not_center*/0, 0, 0, /*variant*/256, /*print*/0);
  void *obj = NULL;
  plink2compressed(plink, plink_t, snps, indiv, f, 15, &obj);
  if (!obj) { fprintf(stderr, "plink2compressed failed: %s\n", mxa_last_error_string()); return 2; }

  int bad = 0;
  for (int trans = 0; trans < 2; trans++) {
    const int n = trans ? 15 : 10, k = trans ? indiv : snps, m = trans ? snps : indiv;
    double *B = malloc(sizeof(double) * (size_t)k * n), *C = malloc(sizeof(double) * (size_t)m * n);
    for (int j = 0; j < n; j++)
      for (int i = 0; i < k; i++) B[(size_t)j * k + i] = -((trans ? 1000.0 : 10.0) * (i + 1) + (j + 1)) * 1e-3;
    dgemm_compressed(trans ? "t" : "n", obj, n, B, k, C, m);
    double maxdiff = 0, maxabs = 0;
    for (int j = 0; j < n; j++)
      for (int r = 0; r < m; r++) {
        long double acc = 0;
        for (int q = 0; q < k; q++) {
          const int s = trans ? r : q, i = trans ? q : r;
          acc += ((long double)Z[(size_t)s * indiv + i] - 2.0L * f[s]) * B[(size_t)j * k + q];
        }
        const double d = fabs((double)acc - C[(size_t)j * m + r]);
        if (d > maxdiff) maxdiff = d;
        if (fabs((double)acc) > maxabs) maxabs = fabs((double)acc);
      }
    printf("dgemm_compressed '%c': %d x %d x %d  max |diff| = %.3e (max |C| = %.3e)\n", trans ? 't' : 'n', snps, indiv, n, maxdiff, maxabs);
    if (!(maxdiff <= 1e-4)) bad = 1;
    free(B); free(C);
  }
  free_compressed(&obj);
  if (obj != NULL) { fprintf(stderr, "handle not cleared\n"); bad = 1; }

  /* integer crossproduct (crossproduct.jl:54-58): indiv x indiv, exact */
  {
    const int ni = indiv < 300 ? indiv : 300;          /* the first ni individuals */
    double *M = malloc(sizeof(double) * (size_t)ni * ni);
    if (snp_multiply_gpu((unsigned char *)plink_t, snps, ni, M, 1) != 0) { fprintf(stderr, "snp_multiply_gpu failed\n"); bad = 1; }
    long wrong = 0;
    for (int a = 0; a < ni; a += 7)
      for (int b = 0; b < ni; b += 5) {
        long acc = 0;
        for (int s = 0; s < snps; s++) acc += (long)Z[(size_t)s * indiv + a] * Z[(size_t)s * indiv + b];
        if (M[(size_t)a * ni + b] != (double)acc) wrong++;
      }
    printf("snp_multiply_gpu: %d x %d, mismatches on the sampled entries: %ld\n", ni, ni, wrong);
    if (wrong) bad = 1;
    free(M);
  }

  /* sparse_times_plink (test_sparse_plink.f90:99; zero-based CSR): 2 sparse rows over the SNPs, result 2 x indiv */
  {
    int ia[3] = {0, 3, 5}, ja[5] = {0, 2, snps - 1, 1, 3};
    double a[5] = {0.5, -1.0, 2.0, 1.5, -0.25}, *C = malloc(sizeof(double) * 2 * (size_t)indiv);
    sparse_times_plink("N", "N", plink, plink_t, snps, indiv, 2, ia, ja, a, C, 2);
    double maxdiff = 0;
    for (int i = 0; i < indiv; i++)
      for (int j = 0; j < 2; j++) {
        double acc = 0;
        for (int t = ia[j]; t < ia[j + 1]; t++) acc += a[t] * Z[(size_t)ja[t] * indiv + i];
        const double d = fabs(acc - C[j + 2 * (size_t)i]);
        if (d > maxdiff) maxdiff = d;
      }
    printf("sparse_times_plink: max |diff| = %.3e\n", maxdiff);
    if (mxa_last_error() || !(maxdiff <= 1e-12)) bad = 1;
    free(C);
  }

  /* solver twin (tests/solve/test.jl): a 3 x 3 Cholesky solve with log-determinant and an upper triangular sparse solve */
  {
    double A[9] = {4, 2, 0, 2, 5, 3, 0, 3, 6}, B[3] = {2, 1, 3}, X[3], logdet = 0;
    int status = 7;
    potrs_solve_gpu(A, 3, B, 1, X, &logdet, 0, &status);
    double r = 0;
    for (int i = 0; i < 3; i++) { double acc = -B[i]; for (int j = 0; j < 3; j++) acc += A[i + 3 * j] * X[j]; if (fabs(acc) > r) r = fabs(acc); }
    printf("potrs_solve_gpu: status %d, residual %.1e, logdet %.12f (log 60 = %.12f)\n", status, r, logdet, log(60.0));
    if (status != 0 || r > 1e-13 || fabs(logdet - log(60.0)) > 1e-12) bad = 1;
    double V[5] = {2, 1, 4, -1, 5}, Xs[3], Bs[3] = {3, 2, 10};      /* U = [2 1 0; 0 4 -1; 0 0 5], one-based COO */
    long I[5] = {1, 1, 2, 2, 3}, J[5] = {1, 2, 2, 3, 3};
    void *sp = NULL;
    sparse2gpu(V, I, J, 5, 3, 1, /*is_lower*/0, &sp, &status);
    if (status != 0 || !sp) { fprintf(stderr, "sparse2gpu failed\n"); bad = 1; }
    else {
      dcsrtrsv_solve_gpu(sp, 'n', Bs, 1, Xs, &status);             /* x = (1, 1, 2) */
      printf("dcsrtrsv_solve_gpu: status %d, x = (%g, %g, %g)\n", status, Xs[0], Xs[1], Xs[2]);
      if (status != 0 || fabs(Xs[0] - 1) > 1e-14 || fabs(Xs[1] - 1) > 1e-14 || fabs(Xs[2] - 2) > 1e-14) bad = 1;
      free_sparse_gpu(&sp, &status);
      if (sp != NULL || status != 0) bad = 1;
    }
  }
  printf(bad ? "FAILED\n" : "c_driver ok\n");
  return bad;
}
