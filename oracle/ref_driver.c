/* TEST INFRASTRUCTURE ONLY -- driver for the reference's own CPU library built by Makefile.ref.
 * Loads oracle/_ref/libmiraculix_ref.so with lazy binding (the unused dense-solver files reference
 * LAPACK routines this image lacks; they are never called on this path, nothing is stubbed) and runs
 *   setOptions_compressed -> plink2compressed -> dgemm_compressed -> free_compressed
 * exactly as the reference's Fortran test does (tests/dgemm_compressed/test_5codesapi.f90:170-195).
 *
 * usage: ref_driver <lib.so> <in.bin> <out.bin> [reps]
 * in.bin  : int32 snps, indiv, n, ldb, ldc, trans(0='N',1='T'), do_not_center, variant, cores
 *           then plink bytes [snps*ceil(indiv/4)], f [snps] f64, B [ldb*n] f64
 * out.bin : C [ldc*n] f64, then f64 seconds of the best dgemm_compressed repetition
 */
#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

typedef void (*setopt_t)(int, int, int, int, int, int, int, int, int, int);
typedef void (*p2c_t)(char *, char *, int, int, double *, int, void **);
typedef void (*dgemm_t)(char *, void *, int, double *, int, double *, int);
typedef void (*free_t)(void **);

static double now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }

int main(int argc, char **argv) {
  if (argc < 4) { fprintf(stderr, "usage: %s lib in out [reps]\n", argv[0]); return 2; }
  int reps = argc > 4 ? atoi(argv[4]) : 1;
  void *lib = dlopen(argv[1], RTLD_LAZY | RTLD_GLOBAL);
  if (!lib) { fprintf(stderr, "dlopen: %s\n", dlerror()); return 3; }
  setopt_t setopt = (setopt_t)dlsym(lib, "setOptions_compressed");
  p2c_t p2c = (p2c_t)dlsym(lib, "plink2compressed");
  dgemm_t dg = (dgemm_t)dlsym(lib, "dgemm_compressed");
  free_t fr = (free_t)dlsym(lib, "free_compressed");
  if (!setopt || !p2c || !dg || !fr) { fprintf(stderr, "missing symbol\n"); return 4; }
  FILE *fi = fopen(argv[2], "rb");
  if (!fi) { perror("in"); return 5; }
  int hdr[9];
  if (fread(hdr, sizeof(int), 9, fi) != 9) return 6;
  int snps = hdr[0], indiv = hdr[1], n = hdr[2], ldb = hdr[3], ldc = hdr[4], trans = hdr[5], nocenter = hdr[6], variant = hdr[7], cores = hdr[8];
  size_t nb = (size_t)snps * ((indiv + 3) / 4);
  char *plink = malloc(nb);
  double *f = malloc(sizeof(double) * snps), *B = malloc(sizeof(double) * (size_t)ldb * n), *C = malloc(sizeof(double) * (size_t)ldc * n);
  if (fread(plink, 1, nb, fi) != nb) return 7;
  if (fread(f, sizeof(double), snps, fi) != (size_t)snps) return 8;
  if (fread(B, sizeof(double), (size_t)ldb * n, fi) != (size_t)ldb * n) return 9;
  fclose(fi);
  for (size_t i = 0; i < (size_t)ldc * n; i++) C[i] = -777.0;
  /* (use_gpu, cores, floatLoop, meanSubstract, ignore_missings, do_not_center, do_normalize, use_miraculix_freq, variant, print) */
  setopt(0, cores, 0, 0, 1, nocenter, 0, 0, variant, 0);
  void *obj = NULL;
  /* the CPU path never reads plink_transposed (5codesChar.cc:368-393); the Fortran benchmark passes plink twice */
  p2c(plink, plink, snps, indiv, f, n, &obj);
  char t = trans ? 'T' : 'N';
  double best = 1e30;
  for (int r = 0; r < reps; r++) {
    double t0 = now();
    dg(&t, obj, n, B, ldb, C, ldc);
    double dt = now() - t0;
    if (dt < best) best = dt;
  }
  FILE *fo = fopen(argv[3], "wb");
  fwrite(C, sizeof(double), (size_t)ldc * n, fo);
  fwrite(&best, sizeof(double), 1, fo);
  fclose(fo);
  /* free_compressed on the CPU path is flagged "will crash?!!!" in the reference (5codesChar.cc:447); skip it */
  (void)fr;
  return 0;
}
