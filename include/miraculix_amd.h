/* miraculix_amd.h -- C ABI of the MI355X-native compressed-genotype GEMM engine.
 *
 * Part 1 is the drop-in boundary: exactly the unmangled C symbols the reference's language bindings
 * bind (dlopen + ccall in src/bindings/Julia/{dgemm_compressed,crossproduct}.jl, bind(C) in src/bindings/Fortran/mod5codesapi.f90).
 * Each declaration cites the reference interface it replaces (paths relative to the reference repo).
 * Part 2 are additive entry points (prefix mxa_) for device-resident operands, SNP-sharded multi-GPU
 * use, on-device .bed staging helpers and measurement; the reference has no counterpart for them.
 *
 * All matrices are column-major fp64.  Plain pointers and sizes only.
 */
#ifndef MIRACULIX_AMD_H
#define MIRACULIX_AMD_H

#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
/* The library is built with -fvisibility=hidden: exactly the entry points declared in this header are exported (no C++ symbol of the
 * implementation reaches the namespace of the Julia / R / Fortran process that loads it). */
#if defined(__GNUC__)
#pragma GCC visibility push(default)
#endif

/* ------------------------------------------------------------------ Part 1: reference ABI */

/* replaces src/miraculix/5codesAPI.c:43-70 (prototype src/miraculix/5codes.h:137-153; doc
 * docs/genotype_matrix_multiplication.md:5-17; Fortran binding src/bindings/Fortran/mod5codesapi.f90:22-40).
 * Process-global options.  This engine is GPU-only: use_gpu == 0 prints a message, sets mxa_last_error() (code 14) and
 * makes every later plink2compressed leave its handle NULL until setOptions_compressed is called again with use_gpu != 0 --
 * the host process (a Julia / R session) is not terminated and no CPU engine is substituted.  The combinations the
 * reference rejects for its GPU path (5codesChar.cc:192-193: use_miraculix_freq != 0, ignore_missings == 0,
 * do_normalize != 0) are fatal (stderr + exit) exactly as there.  cores, floatLoop, meanSubstract, variant have no GPU meaning and are
 * accepted and ignored (src/miraculix/GPUapi.h:38). */
void setOptions_compressed(int use_gpu, int cores, int floatLoop, int meanSubstract, int ignore_missings,
                           int do_not_center, int do_normalize, int use_miraculix_freq, int variant,
                           int print_details);

/* replaces src/miraculix/5codesAPI.c:80-96 -> plink2gpu (src/cuda/dgemm_compressed_cuda.cu:43-170).
 * plink: SNP-major bed payload without the 3 header bytes, snps rows of ceil(indiv/4) bytes;
 * plink_transposed: indiv rows of ceil(snps/4) bytes; f: snps allele frequencies (required when centring);
 * max_n: largest n later passed to dgemm_compressed (buffers grow if exceeded).  The data is copied: the caller
 * may free its buffers afterwards.  Either matrix pointer may also be a device pointer.
 * plink_transposed may be NULL or the same pointer as plink -- the call shape of the reference's CPU path, which never reads it
 * (5codesChar.cc:368-393; utils/benchmark/benchmark.f90:185 passes the same pointer twice): then only the SNP-major matrix is uploaded
 * and the individual-major copy is produced on the device (2-bit transpose of the raw PLINK codes), bit-identical to the object built
 * from two pointers.  On failure *compressed is left NULL and a message is printed (reference: print + handle unset). */
void plink2compressed(char *plink, char *plink_transposed, int snps, int indiv, double *f, int max_n,
                      void **compressed);

/* replaces src/miraculix/5codesAPI.c:98-110 -> dgemm_compressed_gpu (src/cuda/dgemm_compressed_cuda.cu:218-489).
 * trans[0] in {N,n}: C(indiv x n) = (Z - 2*1*f^T) * B(snps x n); {T,t,Y,y}: C(snps x n) = (Z - 2*1*f^T)^T * B(indiv x n);
 * anything else: exit(99) (5codesAPI.c:73-77).  Centring is governed by do_not_center.  Ldb/Ldc are honoured
 * (the reference GPU path ignores them, its CPU path honours them and zero-fills the padding rows of C;
 * so does this).  B and C may each be host or device pointers.  Synchronous. */
void dgemm_compressed(char *trans, void *compressed, int n, double *B, int Ldb, double *C, int Ldc);

/* replaces src/miraculix/5codesAPI.c:159-161 -> freegpu (src/cuda/dgemm_compressed_cuda.cu:176-213).
 * Releases all device memory and sets *compressed = NULL (the reference leaves the caller's pointer dangling). */
void free_compressed(void **compressed);

/* replaces src/miraculix/5codesAPI.c:37-39.  Returns the allele frequencies stored with the object. */
void get_compressed_freq(void *compressed, double *f);

/* replaces src/miraculix/5codesAPI.c:135-157 -> sparseTGenoPlinkApi (5codesChar.cc:472-491) -> sparseTGenoPlink
 * (plinkUint.cc:352-470); Fortran binding src/bindings/Fortran/mod5codesapi.f90:84-100, caller
 * tests/sparse_plink/test_sparse_plink.f90:99.  Sparse (CSR) times packed genotypes, uncentred, missing -> 0.  Behaviour as
 * observed from the reference's library (golden fixtures in tests/golden/sparse_golden.npz): the sparse COLUMN index selects a row
 * of the packed matrix and the result runs over the 2-bit entries of that row --
 *   transcompressed in {N,n}: C (nIdx x indiv, ld Ldc) = S (nIdx x snps)  * Z^T, packed matrix = plink;
 *   transcompressed in {T,t,Y,y}: C (nIdx x snps, ld Ldc) = S (nIdx x indiv) * Z, packed matrix = plink_transposed.
 * rowIdxB (nIdx + 1 entries) / colIdxB / B are ZERO-based CSR; C is zero-filled over Ldc x columns.  transsparse must be N
 * (the reference aborts otherwise; so does this).  Only the packed matrix that is used needs to be non-NULL.  Pointers may be
 * host or device.  Errors: message on stderr, C unwritten, mxa_last_error() != 0. */
void sparse_times_plink(char *transsparse, char *transcompressed, char *plink, char *plink_transposed, int snps, int indiv,
                        int nIdx, int *rowIdxB, int *colIdxB, double *B, double *C, int Ldc);

/* replaces src/miraculix/5codesAPI.c:112-130 (prototype 5codes.h:137-153 region; docs/genotype_matrix_multiplication.md) -> vectorGenoPlinkApi
 * (5codesChar.cc:495-520).  One product straight from the PLINK matrices, no object kept: the DOCUMENTED semantics -- trans in {N,n}: C (indiv x n)
 * = Zc B with B snps x n; {T,t,Y,y}: C (snps x n) = Zc^T B with B indiv x n; f != NULL: centred with the caller's frequencies, f == NULL: uncentred
 * (whatever setOptions_compressed says) -- as the composition plink2compressed + dgemm_compressed + free_compressed of this library.
 * PARITY UNPINNED: in the reference this entry ends in an unconditional BUG abort for every input (f != NULL: 5codesChar.cc:511-513; f == NULL:
 * plink256.cc:332) and nothing binds it, so there is no reference output to compare with; results are checked against the dense oracle.  The reference's
 * "indiv must be a multiple of 32" (5codesChar.cc:510) is not required.  Only the matrix the reference would read needs to be non-NULL ('N':
 * plink_transposed, 'T': plink); when only plink_transposed is given it is transposed on the device first.  Host or device pointers.
 * Errors: message on stderr, C unwritten, mxa_last_error() != 0. */
void dgemm_plink(char *trans, char *plink, char *plink_transposed, int snps, int indiv, double *f, int n, double *B, int Ldb, double *C, int Ldc);

/* replaces src/cuda/snp_multiply_cuda.cu:375-382 (prototype src/cuda/snp_multiply_cuda.h:113-114; Julia binding
 * src/bindings/Julia/crossproduct.jl:54-58, which passes the bool as Cint).
 * ans(indiv x indiv, column-major doubles, full symmetric) = X * X^T where X has `indiv` rows of ceil(snps/4) bytes
 * of 2-bit values; positional meaning as in the reference: `snps` = packed (inner) dimension, `indiv` = output
 * dimension.  is_plink_format applies the reference's byte table first (00->0, 10->1, 11->2, any byte holding a
 * missing 01 pair -> 0xFF).  Exact int32 accumulation.  snp_matrix and ans may be host or device pointers.
 * Returns 0 on success, 1 on failure. */
int snp_multiply_gpu(unsigned char *snp_matrix, int snps, int indiv, double *ans, bool is_plink_format);

/* ---- solver twin (not on the compressed-genotype hot path) behind the reference's solver exports
 * src/cuda/solve_cuda.cu:927-951 (prototypes src/cuda/solve_cuda.h:61-88; Julia binding src/bindings/Julia/solve.jl:45-180;
 * Fortran binding src/bindings/Fortran/modmiraculix_gpu.f90:23-80).  Blocked Cholesky and a synchronisation-free sparse
 * triangular solve written here, Level-3 updates on the fp64 matrix cores included (no vendor library is loaded).  All matrices column-major fp64;
 * pointers may be host or device. */

/* replaces solve_cuda.cu:947-951 -> dense_solve (:70-280): X = A^-1 B by Cholesky (lower triangle of the symmetric positive
 * definite A, input_size x input_size; B, X input_size x rhs_cols) and, if logdet != NULL, *logdet = log det A = sum 2 log L_ii.
 * oversubscribe: 1 = the matrix lives in managed memory (hipMallocManaged), 0 = device memory; anything else is an error.
 * *status = 0 on success, 1 on failure (message on stderr; a matrix that is not positive definite reports the failing minor). */
void potrs_solve_gpu(double *A, unsigned int input_size, double *B, unsigned int rhs_cols, double *X, double *logdet,
                     int oversubscribe, int *status);
/* replaces solve_cuda.cu:927-931: the same, status as the return value */
int potrs_solve(double *A, unsigned int input_size, double *B, unsigned int rhs_cols, double *X, double *logdet,
                int oversubscribe);

/* replaces solve_cuda.cu:933-936 -> sparse_solve_init (:281-578): stage a sparse triangular m x m matrix given as ONE-based COO
 * triplets (V, I, J; 64-bit indices; any order -- they are sorted into CSR here) for solves with exactly `ncol` right-hand
 * sides; is_lower != 0: lower triangular, else upper.  *GPU_obj receives the object (NULL on failure), *status 0 / 1. */
void sparse2gpu(double *V, long *I, long *J, long nnz, long m, long ncol, int is_lower, void **GPU_obj, int *status);
/* replaces solve_cuda.cu:938-941 -> sparse_solve_compute (:709-880): X (m x ncol) = op(A)^-1 B, transA in {N,n}: op(A) = A,
 * {T,t,f}: op(A) = A^T (as the reference); ncol must equal the value given to sparse2gpu. */
void dcsrtrsv_solve_gpu(void *GPU_obj, char transA, double *B, long ncol, double *X, int *status);
/* replaces solve_cuda.cu:943-945 -> sparse_solve_destroy (:580-707): releases the object and sets *GPU_obj = NULL (the Julia
 * test expects a second free to be caught by its NULL check, tests/solve/test.jl:129). */
void free_sparse_gpu(void **GPU_obj, int *status);

/* ------------------------------------------------------------------ Part 2: additive entry points */

/* status of the most recent fallible API call of the process: 0 = it succeeded (every such entry clears the status first);
 * the message is valid until the next call. */
int mxa_last_error(void);
const char *mxa_last_error_string(void);

/* number of visible HIP devices, or -1 if the runtime cannot be initialised */
int mxa_device_count(void);

/* SNP-sharded staging for one-process-per-GPU use: like plink2compressed, but this object holds only SNPs
 * [snp_begin, snp_end) of the full matrix.  plink points at the FULL SNP-major payload (row pitch ceil(indiv/4));
 * plink_transposed at the FULL individual-major payload (row pitch ceil(snps_total/4)); f at the FULL frequency
 * vector.  snp_begin must be a multiple of 4 so that packed bytes split cleanly.
 * For 'N' the caller passes rows [snp_begin, snp_end) of B and sum-reduces C over the shards
 * (the centring term is a partial sum too and rides along); for 'T' C holds rows [snp_begin, snp_end).
 * plink_transposed may be NULL (or == plink): the shard transposes its own SNP block on the device. */
void mxa_plink2compressed_shard(char *plink, char *plink_transposed, int snps_total, int indiv, int snp_begin,
                                int snp_end, double *f, int max_n, void **compressed);

/* dgemm_compressed with 64-bit leading dimensions on an explicit HIP stream (NULL = the object's own stream),
 * device pointers only, asynchronous when sync == 0.  Returns 0 / 1.
 * Asynchronous for EVERY n under the default engine (round 5): the verdict of the exact int8 route of narrow products and peeled columns is formed on
 * the device; nothing is read back.  (The first call of a new shape on an object may grow its workspace, which waits for the stream once; the opt-in
 * engine `i8-exact` reads three integers per call, as documented with it.)
 * ONE CALL IN FLIGHT PER OBJECT: every multiply on an object uses that object's workspace (fragment-ordered B, split-K
 * partials, column sums).  Calls on the same object must therefore be serialised on ONE stream (or the caller must wait
 * for the previous call before issuing the next on another stream); this includes mxa_gram_matvec and dgemm_compressed.
 * Different objects are independent.  Not available on multi-device objects (MIRACULIX_NUM_GPUS > 1). */
int mxa_dgemm_compressed_device(char trans, void *compressed, int n, const double *dB, long ldb, double *dC,
                                long ldc, void *hip_stream, int sync);

/* one step of the GRM-based solvers (reference loop: examples/iterative_solver/grm_solve_cg.jl:74-84, which calls
 * dgemm_compressed 'T' then 'N' and notes the cost of moving the operands each time at dgemm_compressed_cuda.cu:251-252):
 * out (indiv x n, ld ldo) = Zc * (Zc^T * V), V indiv x n (ld ldv); the snps x n intermediate stays in HBM.  Centring as set by
 * setOptions_compressed.  V / out host or device.  On a SNP shard the result is that shard's partial sum.  Returns 0 / 1. */
int mxa_gram_matvec(void *compressed, int n, const double *V, long ldv, double *out, long ldo);
/* the same with device-resident V / out (memory of the object's device) and optional asynchrony: sync == 0 returns when both products are
 * enqueued on the object's stream -- a blocking stream, so work the caller enqueues afterwards on the device's default stream (PyTorch, hipBLAS
 * on stream 0) is ordered behind it and a CG / GBLUP loop on device-resident vectors never waits on the host: the ~40 us between two
 * synchronous calls (return, caller, next launch) disappear from every iteration.  Single-device objects only.  Returns 0 / 1.
 * (No n waits on the host under the default engine: see mxa_dgemm_compressed_device.) */
int mxa_gram_matvec_device(void *compressed, int n, const double *dV, long ldv, double *dOut, long ldo, int sync);

/* on-device .bed staging helpers (reference counterparts live in the bindings:
 * transpose_genotype_matrix src/bindings/Julia/compressed_operations.jl:45-66, popcount frequencies
 * src/bindings/Julia/read_plink.jl:199-203).  Pointers may be host or device. */
int mxa_transpose_2bit(const unsigned char *in, long rows, long cols, unsigned char *out);
/* f_s = (sum of the allele counts of SNP s) / (2 indiv) with the decode the multiply uses: 00 -> 0, 10 -> 1, 11 -> 2 and the
 * missing code 01 -> 0, so that f is exactly the column mean / 2 of the matrix dgemm_compressed multiplies with.
 * DEVIATION on data with missing genotypes: the reference binding counts set bits (read_plink.jl:199-203), i.e. a missing
 * 01 adds 1; on missing-free data (the only data the reference's tests and crossproduct accept, read_plink.jl:213) both agree. */
int mxa_allele_freq(const unsigned char *plink, long snps, long indiv, double *f);

/* PLINK .bed staging owned by the library: reads the SNP-major .bed file `bed_path` (3-byte magic 6c 1b 01, then snps rows of
 * ceil(indiv/4) bytes; reference reader: src/bindings/Julia/read_plink.jl:161-222), uploads it in chunks, builds the
 * individual-major copy with the on-device 2-bit transpose and the allele frequencies with the on-device popcount
 * (f_s = allele count / (2 indiv)), and returns the same kind of object as plink2compressed.  snps / indiv <= 0: taken
 * from the line counts of the .bim / .fam files next to the .bed.  f_out (optional, host, snps doubles) receives the
 * frequencies; snps_out / indiv_out (optional) the dimensions.  Returns 0 / 1; *compressed is NULL on failure. */
int mxa_bed2compressed(const char *bed_path, int snps, int indiv, int max_n, void **compressed, double *f_out, int *snps_out,
                       int *indiv_out);

/* The same for SNP rows [snp_begin, snp_end) of the file only: ONLY those rows are read (fseek), the individual-major block and
 * the frequencies of the range are produced on the device, and the object behaves like one made by mxa_plink2compressed_shard --
 * without any process ever holding the full SNP-major or individual-major matrix (one-process-per-GPU jobs: every rank stages
 * its own range; the in-process sharder below does the same per device).  snps / indiv <= 0: from .bim / .fam.  f_out
 * (optional, host) receives the snp_end - snp_begin frequencies of the range.  Returns 0 / 1. */
int mxa_bed2compressed_range(const char *bed_path, int snps, int indiv, int snp_begin, int snp_end, int max_n, void **compressed,
                             double *f_out);

/* Incremental staging (round 5): plink2compressed (5codesAPI.c:80-96 -> plink2gpu, dgemm_compressed_cuda.cu:43-170) wants the whole PLINK matrix
 * behind one pointer and the reference gives up when matrix + object exceed the device (dgemm_compressed_cuda.cu:93-100).  Here the object is
 * allocated first -- ONE packed copy (a single-orientation object, see mxa_single_orientation below) -- and filled by blocks of SNP rows, so that
 * BASELINE config 4 at its full 5M x 200k (250 GB packed) is staged on one 288 GB device from a generator / reader that holds one block at a time.
 *   mxa_plink2compressed_begin: allocate (packed matrix zeroed; options as plink2compressed).  *compressed NULL on failure.
 *   mxa_plink2compressed_rows : SNP rows [snp_begin, snp_begin + nrows) = nrows compact PLINK rows of ceil(indiv/4) bytes, host or device memory;
 *                               blocks may arrive in any order and must cover every row once.  f_rows: the nrows allele frequencies of the block,
 *                               or NULL -- then they are counted on the device with mxa_allele_freq's rule.  Returns when the block buffer may be reused.
 *   mxa_plink2compressed_end  : seal; get_compressed_freq returns the frequencies.  Products are refused (error 19) before this call.
 * mxa_bed2compressed(_range) use the same path for single-orientation objects: the .bed is streamed into the object in 64 MB chunks.
 * Products whose split-K partial sums would not fit beside such an object run their K splits in groups with the running sum kept in C
 * (same pieces, same order of additions: bit-identical to the one-pass product).  Return 0 / 1. */
int mxa_plink2compressed_begin(long snps, long indiv, int max_n, void **compressed);
int mxa_plink2compressed_rows(void *compressed, const unsigned char *plink_rows, long snp_begin, long nrows, const double *f_rows);
int mxa_plink2compressed_end(void *compressed);

/* ---- several GPUs behind the reference ABI (single process).  With MIRACULIX_NUM_GPUS=G (> 1) in the environment,
 * plink2compressed and mxa_bed2compressed return ONE handle that owns G per-device objects over contiguous SNP blocks
 * (boundaries at multiples of 4; devices HIP_DEVICE/CUDA_DEVICE (default 0) + 0..G-1 modulo the visible device count -- more
 * shards than devices puts several blocks on one GPU).  dgemm_compressed / mxa_gram_matvec / get_compressed_freq /
 * free_compressed accept it unchanged: 'N' multiplies every block on its own device (one host thread per device, so host
 * operands travel over every GPU's own PCIe link side by side) and sum-reduces the indiv x n partials onto the first device --
 * by default with peer-to-peer pushes over xGMI and ONE addition kernel in ascending block order (bitwise reproducible),
 * with MXA_REDUCE=rccl by ncclReduce (RCCL is dlopen()ed; needs distinct devices); 'T' writes disjoint row blocks, no exchange.
 * B / C may be host memory or memory of any of the devices (work the caller enqueued on that device's default stream is waited for
 * first).  mxa_dgemm_compressed_device is not available on such a handle; mxa_dgemm_compressed_multi (below) takes per-device operand slices.
 * Peer access between the devices is enabled at creation and the verdicts are logged (PRINT_LEVEL > 0) and reported by mxa_multi_shard_info.
 * NOT YET MEASURED on more than one physical GPU: all of it runs in the tests with several shards on one device.
 * snp_multiply_gpu with host operands follows the same variable: device g computes the column panel [c_g, c_g+1) of the symmetric result (equal
 * numbers of 256-column tiles = equal work), staging the packed matrix itself, and downloads it over its own PCIe link into its slab of the host
 * matrix -- independent units, no exchange; bit-identical to the single-device result.
 * mxa_num_shards: number of per-device objects behind a handle (1 for an ordinary one).  mxa_shard_bounds: block g of the
 * partition of `snps` into `shards` blocks; returns the number of non-empty blocks. */
int mxa_num_shards(void *compressed);
int mxa_shard_bounds(long snps, int shards, int g, long *begin, long *end);

/* dgemm_compressed on a multi-device object with the operands handed over PER SHARD, so that nothing but the indiv x n partial sums
 * crosses a device boundary (with ONE B / C pointer, as through dgemm_compressed, the device that holds them is a hub: every shard
 * copies its slice from / to it).  Arrays of mxa_num_shards(compressed) pointers; block g = mxa_shard_bounds(snps, shards, g, ..):
 *   'N': B_per_shard[g] = rows [begin_g, end_g) of B (leading dimension ldb), normally memory of shard g's device;
 *        C_per_shard[0] receives the reduced indiv x n result (leading dimension ldc; memory of the first shard's device, of any other
 *        device, or of the host); the other entries of C_per_shard are ignored.
 *   'T': B_per_shard[g] = the whole indiv x n matrix B as shard g reads it (leading dimension ldb; a NULL entry g > 0 makes shard g read
 *        B_per_shard[0] across devices); C_per_shard[g] = rows [begin_g, end_g) of C (leading dimension ldc >= end_g - begin_g; rows
 *        beyond the block are not touched).
 * sync == 0: the call returns when the work is enqueued (device operands only; a host operand makes the call synchronous).  Products
 * issued back to back on one object are ordered like calls on one stream, but the transfers of an 'N' product's partial sums and their
 * addition on the first device run BESIDE the next product ('T' of the same step) on copy streams.  mxa_multi_synchronize() waits for
 * everything issued on the object (a later product that READS or overwrites the memory the previous 'N' product delivered its result to is
 * ordered behind that delivery; products on unrelated memory are not held up).  Results are those of dgemm_compressed on the same object (same
 * kernels, same fixed-order reduction).
 * Reference counterpart of the need: src/cuda/dgemm_compressed_cuda.cu:251-252 (operands cross PCIe on every call).  Returns 0 / 1. */
int mxa_dgemm_compressed_multi(char trans, void *compressed, int n, const double *const *B_per_shard, long ldb,
                               double *const *C_per_shard, long ldc, int sync);
int mxa_multi_synchronize(void *compressed);

/* reduction of the 'N' partial sums on a multi-device object: 0 = peer-to-peer pushes + ONE addition kernel in ascending shard order
 * (default; bitwise reproducible), 1 = RCCL ncclReduce (one rank per shard, all in this process; needs one device per shard).
 * Returns 0, 1 (error) or 2 (RCCL not applicable: several shards share a device; the reduction is unchanged).  The FIRST RCCL reduction of
 * an object is cross-checked: the same partial sums are also reduced peer-to-peer and the two results compared (<= 1e-13 of the largest
 * entry, else the product fails with error 18); mxa_multi_get_info reports the difference. */
int mxa_multi_set_reduction(void *compressed, int kind);

/* what a multi-device object is made of and what it has done since mxa_multi_reset_profile (HIP events on the streams the work ran on) */
typedef struct mxa_multi_info {
  int shards, devices, root_device;
  int reduction;                     /* 0 peer-to-peer fixed order, 1 RCCL */
  int reductions; double reduce_ms;  /* addition kernel on the root device (RCCL: the ld-padded copy of the received sum) */
  int rccl_checked; double rccl_vs_p2p_max_rel_diff;   /* -1 until an RCCL reduction has been cross-checked */
} mxa_multi_info;
typedef struct mxa_shard_info {
  int device; long snp_begin, snp_end;
  int peer_to_root, peer_from_root;          /* 1 peer access enabled (direct xGMI), 0 not available (copies are staged through the host), -1 same device */
  int kernel_launches; double kernel_ms;     /* dominant kernel of this shard's products */
  int in_copies; double in_ms;               /* operand distribution: copies of a B that did not live on this shard's device */
  int out_copies; double out_ms;             /* result gather: copies of a 'T' row block to a C that did not live on this shard's device */
  int pushes; double push_ms;                /* partial sums pushed to the root device (RCCL: shard 0 = the ncclReduce) */
} mxa_shard_info;
int mxa_multi_get_info(void *compressed, mxa_multi_info *out);
int mxa_multi_shard_info(void *compressed, int shard, mxa_shard_info *out);
int mxa_multi_reset_profile(void *compressed);

/* Output-tile sharding of the crossproduct for one-process-per-GPU use (packed matrix replicated, independent
 * units, no collective): columns [col_begin, col_end) of the symmetric result of snp_multiply_gpu, i.e. the contiguous slab
 * ans + col_begin*indiv of the full column-major matrix, into `panel` (indiv rows, col_end - col_begin columns, leading
 * dimension ld).  col_begin must be a multiple of 256, col_end a multiple of 256 or == indiv.  upper_only != 0 computes only
 * rows [0, col_end) of the panel, i.e. everything above its diagonal block and the block itself (a host panel gets zeros in the
 * rows below, a device panel is left untouched there) -- half the total work for callers that exchange the transposed blocks
 * (miraculix_amd/distributed.py: crossprod_sharded).  Same argument meaning otherwise as snp_multiply_gpu.  Returns 0 / 1. */
int mxa_snp_multiply_panel(const unsigned char *snp_matrix, int snps, int indiv, int col_begin, int col_end, int upper_only,
                           double *panel, long ld, int is_plink_format);

/* GRM and LD with the post-processing done on the device before the result leaves HBM (reference: host BLAS in
 * src/bindings/Julia/crossproduct.jl:83-110 grm(), :128-152 ld(); maths docs/grm.md:5-12).
 * mxa_grm: G(indiv x indiv) = P Z Z^T P^T [/ (2 sum f(1-f))], plink_transposed = indiv rows of ceil(snps/4) bytes.
 * mxa_ld : R(snps x snps)  = D^-1/2 (Z^T Z - 4 indiv f f^T) D^-1/2,  plink = snps rows of ceil(indiv/4) bytes.
 * Pointers may be host or device.  Return 0 / 1.
 * The element-wise map runs INSIDE the crossproduct epilogue (round 3): the column sums and the diagonal of the crossproduct are formed from
 * the packed matrix before the product (exact integers), so the result is written once and a host result leaves through the same slab
 * pipeline as snp_multiply_gpu's.  The reference's divisions (by 2 sum f(1-f); by sigma_i, sigma_j) are multiplications by reciprocals
 * (<= 1 ulp from the quotients).  MXA_XPROD_FUSED_POST=0 runs the three separate passes over the result instead (bit-identical). */
int mxa_grm(const unsigned char *plink_transposed, int snps, int indiv, double *G, int is_plink_format, int do_scale,
            const double *allele_freq);
int mxa_ld(const unsigned char *plink, int snps, int indiv, double *R, int is_plink_format, const double *allele_freq);

/* multiply engine of dgemm_compressed (process-wide; MXA_ENGINE in the environment sets the initial one).  Details and error bounds: DESIGN.md 3.2 / 3.3.
 *
 *   id  MXA_ENGINE   arithmetic                                                                     host waits
 *   0   (default)    n >= 7: fp64 MFMA, the reference's FMA chains; n <= 6 and the 1-3 odd columns   never
 *                    of n = 4q + r: exact int8 slicing of B when a device-side check proves it
 *                    exact (|err| <= 3.02 (S-1) 2^-53 sum|z b|), else fp64
 *   1   i8           int8 slicing for every n, 7 digits per column, no exactness check               never
 *   3   f64-strict   fp64 for every n (n <= 2: pair tables)                                          never
 *   4   i8-exact     int8 slicing for every n with the digit count chosen per call so that B is      once per call
 *                    represented without error (S <= 24; otherwise engine 0's path)
 *
 * Only engine 0 is ever the benchmark's `value`.  Ids 2 and 5 (small-n-i8, i8-guarded; rounds 3-5) are retired.
 * mxa_set_engine returns the previous id; an invalid id leaves the engine unchanged.
 * mxa_last_path: kernel family of the MAIN part of the most recent product (the 4q columns of a peeled n = 4q + r; the peeled columns carry their own
 * device-side verdict, which is not reported): 0 = k_gemm, 1 = k_lut (fp64 pair tables), 2 = k_gemm_i8 / k_gemm_i8_tn, 3 = fp64 behind a declined
 * exactness check (read from the device when this is called). */
int mxa_set_engine(int engine);
int mxa_get_engine(void);
int mxa_last_path(void);
/* Range of the fp64 MFMA path.  k_gemm feeds the genotype operand as the denormal double z * 2^-1074 and scales every column of B by a
 * power of two so that its largest |entry| sits just below 2^900; products of entries up to 847 binades below their column's largest
 * one are then normal doubles and every result equals the plain fp64 FMA chain of the reference
 * (src/cuda/dgemm_compressed_cuda.h:259-266) bit for bit.  A per-call check on the device finds columns whose non-zero entries span more
 * than 800 binades (about 240 decades) or hold inf / NaN; the product is then redone with plain fp64 operands (v_cvt_f64_u32, no
 * scaling) -- same arithmetic as the reference for every input, at twice the time for such calls.  mxa_last_range_fallback: 1 if the
 * most recent fp64-MFMA product on this (single-device) object took that fallback, 0 if not, -1 if unknown. */
int mxa_last_range_fallback(void *compressed);

/* measurement: HIP-event timing of the dominant kernel on the stream it is launched on.
 * mxa_profile_reset() clears the counters; after some dgemm_compressed / snp_multiply_gpu calls
 * mxa_profile_get() returns the number of dominant-kernel launches and their summed duration in ms. */
void mxa_profile_reset(void);
void mxa_profile_get(int *launches, double *total_ms);

/* geometry of the last dgemm_compressed call (for roofline accounting): rows, inner dim, n, split count */
void mxa_last_geometry(long *m, long *k, int *n, int *splits, int *a_tile, int *c_tile);
/* doubles of split-K partial sums the fp64 MFMA launch of an m x k x n product writes (no device needed: tests check that the plan of the
 * columns left after a peel, which can be LARGER than the plan of all n columns, never outgrows the workspace) */
long mxa_plan_partial_doubles(long m, long k, int n);
/* One packed copy per object (round 5 default; a property of the object from plink2compressed / mxa_bed2compressed / mxa_plink2compressed_shard /
 * mxa_plink2compressed_begin on).  Only the SNP-major copy is stored -- half the HBM (config 5 at its full 2M x 100k: 50 GB instead of 100; config 4 at its
 * full 5M x 200k: 250 GB, on one device) and half the staging upload; plink_transposed is not read.  Both products read that one copy: 'T' in the plain
 * form of k_gemm (output rows = packed rows), 'N' in the transposed-operand form (output rows = packed columns); since round 5 the two forms share one
 * permuted K order and run at the same rate (0.957-0.962 of the fp64 MFMA peak at BASELINE config 2; bit-identical results).  The exact int8 route of
 * n <= 6 and of peeled columns: 'T' on k_gemm_i8, 'N' on k_gemm_i8_tn (n <= 3: one digit tile, a CG step within 2 % of a two-copy object's; n = 4..6: two
 * tiles in ONE pass over the matrix, 1.44-1.50 ms against 1.3 on 500k x 50k; the fp64 tile would take 3.2-4.4).
 * MXA_SINGLE_ORIENTATION=0 in the environment of the creating call asks for BOTH copies (what rounds 1-4 stored): 'N' then runs the plain kernels
 * everywhere (4 <= n <= 6: 10-15 % faster), and the opt-in engines i8 / i8-exact multiply 'N' at wide n on the plain int8 kernel (on a one-copy object: on the transposed-operand kernel in column chunks of at most six digit tiles, one pass over the packed matrix per two tiles).  If the two copies do not fit the
 * device's free memory and one does, one is kept and a line on stderr says so (the reference reports "Not enough device memory" there,
 * cuda_utils.cu:162-185); a multi-device object decides once for all its shards.  Results of one-copy and two-copy objects agree to rounding
 * (bit-identical on the fp64 MFMA path and for integer-valued operands).  mxa_single_orientation: 1 / 0 (multi-device object: of its shards), -1 for an
 * invalid handle. */
int mxa_single_orientation(void *compressed);
/* capacity (doubles) of the partial-sum workspace an object holds right now; -1 for an invalid / multi-device object */
long mxa_partial_capacity(void *compressed);

#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* MIRACULIX_AMD_H */
