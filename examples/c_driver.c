/* c_driver.c -- the reference's Fortran integration test (tests/dgemm_compressed/test_5codesapi.f90:170-210 and its
 * '_t' twin) restated as a plain C program against the C ABI: synthetic PLINK data, ncol = 10 ('n') / 15 ('t'),
 * B = -(10 i + j) resp. -(1000 i + j), centred by f, result compared with a dense matmul on the decoded genotypes,
 * failure if any |difference| > 1e-4 (the reference's own acceptance threshold; observed ~1e-10).
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
  printf(bad ? "FAILED\n" : "c_driver ok\n");
  return bad;
}
