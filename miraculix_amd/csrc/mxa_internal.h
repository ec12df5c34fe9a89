// mxa_internal.h -- private types shared by the host side (api/staging) and the kernel launchers.
#pragma once
#include <hip/hip_runtime.h>
#include <cstddef>
#include <cstdint>
#include <functional>
#include <utility>
#include <vector>
#include "mxa_plan.h"

namespace mxa {

// ---- geometry constants of the fp64 MFMA kernel (see DESIGN.md "dgemm kernel")
constexpr int kWaves = 4;          // waves per workgroup
constexpr int kKStep = 16;         // genotypes consumed by one v_mfma_f64_4x4x4_4b_f64 (4 blocks x K=4)
constexpr int kSlabSteps = 8;      // K-steps per LDS slab
constexpr int kSlabK = kKStep * kSlabSteps;   // 128 genotypes = 32 packed bytes per row per slab
constexpr int kSlabBytes = kSlabK / 4;
constexpr int kRowAlign = 512;     // packed matrices are padded to a multiple of this many rows (two tiles: the lookup kernel's row block)
constexpr int kTileRows = 256;     // rows per HBM tile
constexpr int kTileBytes = kTileRows * kSlabBytes;   // 8 KiB

struct Options {
  bool gpu = true;
  bool centered = true;      // genetics.centered = !do_not_center (5codesAPI.c:59-68)
  int print_level = 0;       // env PRINT_LEVEL / print_details
  bool set = false;
};
Options &options();
int env_print_level();

// One packed, recoded genotype matrix resident in HBM: `rows` rows of `k` genotypes, 2 bits each holding the
// allele count z in {0,1,2} (PLINK code 01 "missing" already mapped to 0), zero padded to rows_pad x k_pad.
// TILED layout: the matrix is cut into tiles of kTileRows rows x one slab (kSlabK genotypes = kSlabBytes bytes per
// row); a tile is stored as kTileRows consecutive 32-byte row pieces (8 KiB), tiles of one row block are consecutive
// along K:   byte b of row r  ->  ((r / 256) * nslabs + b / 32) * 8192 + (r % 256) * 32 + b % 32.
// Every slab a workgroup needs is then one contiguous 4-8 KiB run, so the LDS-DMA is lane-linear on both sides and HBM is
// read in whole lines exactly once.  `pitch` = nslabs * kSlabBytes is the logical row length in bytes.
struct PackedMatrix {
  uint8_t *d = nullptr;
  long rows = 0, k = 0;
  long rows_pad = 0, k_pad = 0;
  size_t pitch = 0;
  long nslabs = 0;
};

struct Workspace {
  double *d_Bstage = nullptr; size_t cap_Bstage = 0;   // host B staged here
  double *d_Cstage = nullptr; size_t cap_Cstage = 0;   // result staged here when C is a host pointer
  double *d_Bp = nullptr;     size_t cap_Bp = 0;       // B in MFMA fragment order
  double *d_P = nullptr;      size_t cap_P = 0;        // split-K partial slabs
  size_t big_budget = 0;                               // doubles of partial sums a product beyond 4 GiB may hold; decided once per object (partial_budget)
  double *d_colpart = nullptr; size_t cap_colpart = 0; // column-sum partials + sums
  double *d_tmp = nullptr;    size_t cap_tmp = 0;      // snps x n intermediate of mxa_gram_matvec
  int *d_exp = nullptr;       size_t cap_exp = 0;      // per-column exponents of B (denormal-operand mode)
  int *d_denflag = nullptr;                            // [0] != 0: a column of B spans more binades than the denormal-operand mode carries (kDenMaxSpan); [1]: guard of the int8 route
  int *d_ctr = nullptr; unsigned ctr_next = 0;         // 16 blocks of 16 ints: work-queue counters of k_gemm launches, handed out in turn (next_ctr)
  void *d_i8 = nullptr;       size_t cap_i8 = 0;       // int8 engine: exponents, slices of B, int32 partials (bytes)
};

constexpr uint32_t kMagic = 0x4d584131u;  // "MXA1"
constexpr uint32_t kMagicMulti = 0x4d58414du;  // "MXAM": SNP-sharded object over several devices (mxa_multi.cpp)

// what one object has done since the last reset: the dominant kernel (HIP events on the stream it ran on) and, when an operand did not
// live on the object's device, the staging copies in (B) and out (C)
struct ObjectProfile {
  int launches = 0;  double kernel_ms = 0.0;
  int in_copies = 0; double in_ms = 0.0;
  int out_copies = 0; double out_ms = 0.0;
};

struct Handle {
  uint32_t magic = kMagic;
  int device = 0;
  long snps = 0, indiv = 0;
  PackedMatrix snp_major;    // rows = snps,  k = indiv
  PackedMatrix ind_major;    // rows = indiv, k = snps.  Every product can be computed from either copy (mxa_api.cpp: gemm_use_tr).
  // Single-orientation object (MXA_SINGLE_ORIENTATION=1 at plink2compressed, or by itself when two copies do not fit the device; round 4): only the SNP-major copy exists -- half the HBM, half the staging.
  // ind_major then holds the DIMENSIONS of the missing copy (for the launch plans) with d == nullptr; 'T' products run in the plain form, 'N' products in the
  // transposed-operand forms (k_gemm<..., TR>, k_gemm_i8_tn), both on snp_major.
  bool single = false;
  // incremental staging (mxa_plink2compressed_begin / _rows / _end, round 5): the object exists, its packed matrix is being filled by SNP-row blocks;
  // products are refused until _end
  bool staging = false;
  long staged_rows = 0;
  std::vector<std::pair<long, long>> staged_iv;   // disjoint [begin, end) blocks appended so far, sorted
  double *d_f = nullptr;     // snps
  double *h_f = nullptr;
  bool has_f = false;
  int max_n = 0;
  Workspace ws;
  hipStream_t stream = nullptr;
  // timing of the dominant kernel: two event pairs used alternately (created once, not per call), so that reading the pair of the product
  // before last never waits for work that was just enqueued (asynchronous callers keep two products in flight)
  hipEvent_t ev0[2] = {nullptr, nullptr}, ev1[2] = {nullptr, nullptr};
  bool prof_pending[2] = {false, false};     // the pair was recorded and has not been read yet
  int prof_slot = 0;                         // pair the next product records
  hipEvent_t ev_in[2] = {nullptr, nullptr}, ev_out[2] = {nullptr, nullptr};   // around the staging copies of a non-local B / C
  bool in_pending = false, out_pending = false;
  ObjectProfile prof;                        // per object, next to the process-wide profile()
  double prof_last_kernel_ms = 0.0, prof_last_in_ms = 0.0, prof_last_out_ms = 0.0;   // the most recent harvested durations (phase clock line)
  // host-operand pipeline (mxa_api.cpp: gemm_host_pipelined): two compute streams for alternating chunks, events per chunk
  hipStream_t pipe[2] = {nullptr, nullptr};
  hipEvent_t pev[18] = {};                   // [0] start, [1] operands ready, [2..9] upload of chunk c, [10..17] chunk c computed
};

inline int *next_ctr(Workspace &w) { return w.d_ctr + 16 * (w.ctr_next++ & 15); }

struct Profile {
  int launches = 0;
  double total_ms = 0.0;
};
Profile &profile();

// path: 0 k_gemm, 1 k_lut, 2 k_gemm_i8, 4 decided on the device (flag != 0: k_lut, else k_gemm_i8; resolved by mxa_last_path())
struct Geometry { long m = 0, k = 0; int n = 0, splits = 0, a = 0, c = 0, path = 0; const int *d_flag = nullptr; int flag_dev = 0; };
Geometry &last_geometry();

// ---- host-side engine entry points shared by mxa_api.cpp (single device) and mxa_multi.cpp (SNP shards over several devices)
// device < 0: taken from HIP_DEVICE / CUDA_DEVICE / the current device
int create_handle(const uint8_t *plink, size_t plink_pitch, const uint8_t *plink_t, size_t plink_t_pitch, long snps, long indiv,
                  const double *f, int max_n, void **out, int device = -1);
// MXA_SINGLE_ORIENTATION (round 5): unset / 1 = the SNP-major copy alone (default), 0 = both copies while they fit the device's free memory (returned as
// policy 2: if they do not fit and one copy does, one copy is kept and a line on stderr says so).  mxa_multi.cpp decides once per object (all shards alike)
// and tells its workers' create_handle calls through tl_single_override (-1: decide here).
int single_orientation_policy();
size_t object_footprint(long snps, long indiv, int max_n, bool single);   // device bytes of a staged object: packed copies (tile padding included) + workspace
extern thread_local int tl_single_override;
void destroy_handle(Handle *h);
// incremental staging of a single-orientation object: allocate (packed SNP-major matrix zeroed, workspace), then SNP-row blocks in any order, then seal.
// rows: compact PLINK rows (pitch ceil(indiv/4)), host or device; f_rows nullable: the block's frequencies are then counted on the device (k_allele_freq).
int begin_handle(long snps, long indiv, int max_n, void **out, int device = -1);
int append_rows(Handle *h, const uint8_t *rows, long snp_begin, long nrows, const double *f_rows);
int end_handle(Handle *h);
// pointer classification: 0 = host, 1 = device / managed memory (its device in *dev)
int ptr_location(const void *p, int *dev);
// One product on one device object, operands anywhere: B (k x n, ld ldb) and C (m x n, ld ldc) may be host memory, memory of this
// object's device, or memory of another device (then they are staged through this object's buffers).  fill_rows (>= m): rows
// [m, fill_rows) of every column of C are written as zeros (the plain ABI passes ldc, like the reference's CPU path; a SNP shard
// of a 'T' product passes its own row count so that it does not touch its neighbours' rows).  Asynchronous on the object's stream
// unless sync; with timing the caller reads the events later with harvest_profile().
int gemm_any(Handle *h, bool trans, int n, const double *B, long ldb, double *C, long ldc, long fill_rows, bool sync, bool timing);
int gram_any(Handle *h, int n, const double *V, long ldv, double *out, long ldo, bool sync);
void harvest_profile(Handle *h);
// B / V lives in the memory of device `src_dev` != the current one: whatever the caller enqueued on that device's default stream (the
// producer of the operand, e.g. a PyTorch op) must be complete before another device's stream reads it -- a blocking stream orders
// itself against the legacy default stream of ITS OWN device only.  Host-blocking; restores the current device.
int sync_foreign_producer(int src_dev);
// peer access cur -> peer: 1 enabled (now or before), 0 not available; errors other than "already enabled" are reported as 0
int enable_peer(int cur, int peer);
// called once by every shard worker thread: this thread issues transfers at the same time as other threads
void mark_thread_concurrent();
int select_device();                       // honours HIP_DEVICE / CUDA_DEVICE; -1 + error when no device
// SNP-sharded objects (mxa_multi.cpp)
int multi_requested();                     // MIRACULIX_NUM_GPUS (>= 1)
int multi_create(const uint8_t *plink, const uint8_t *plink_t, long snps, long indiv, const double *f, int max_n, int shards, void **out);
int multi_create_from_bed(const char *base, long snps, long indiv, int max_n, int shards, void **out, double *f_out);
int multi_gemm(void *obj, bool trans, int n, const double *B, long ldb, double *C, long ldc);
int multi_gram(void *obj, int n, const double *V, long ldv, double *out, long ldo);
int multi_single(const void *obj);         // 1: the shards keep the SNP-major copy only, 0: both copies (all shards of an object are alike)
void multi_freq(void *obj, double *f);
void multi_destroy(void *obj);
bool is_multi(const void *obj);
// run job(g, device) for g = 0 .. parts-1 at the same time, each in a pooled worker thread bound to device (first device + g) modulo the
// visible device count; returns the OR of the jobs' return codes.  The caller's current device is restored.
int run_on_devices(int parts, const std::function<int(int, int)> &job);
// one SNP range of a .bed file -> object on `device` (rows [snp_begin, snp_end) are read, transposed and counted on the device)
int bed_range_to_handle(const char *base, long snps_total, long indiv, long snp_begin, long snp_end, int max_n, int device, void **out,
                        double *f_out_local);
// C[r + j ldc] = sum_g parts[g][r + j m] (ascending g: fixed order), rows [m, fill_rows) zero
constexpr int kMaxShards = 64;
struct PartList { const double *p[kMaxShards]; int count; };
int launch_reduce_parts(const PartList &parts, long m, int n, double *dC, long ldc, long fill_rows, hipStream_t s);

// ---- error handling: print + remember (reference: cuda_utils.cu:83-90 prints "Internal error in function ...")
void set_error(int code, const char *fmt, ...);
void clear_error();   // every fallible API entry starts with it: mxa_last_error() describes the most recent such call
bool check_hip(hipError_t e, const char *func, int line);
#define MXA_HIP(x) do { if (!::mxa::check_hip((x), __func__, __LINE__)) return 1; } while (0)
void debug_info(const char *fmt, ...);
// Wall-clock phases of ONE ABI call (host side), printed as one line under PRINT_LEVEL > 0 / print_details: where the time of a call through the plain
// symbols goes -- operand classification, workspace growth, the staged upload of a pageable B (the host thread blocks in it), the enqueueing of the
// kernels, the final stream wait.  Off: one branch per mark.  gemm_any arms tl_call_clock, gemm_device marks through it.
struct CallClock {
  bool on = false;
  double t0 = 0.0, last = 0.0;
  char line[512]; int len = 0;
  static double now();
  void start(bool enabled) { on = enabled; len = 0; line[0] = 0; if (on) t0 = last = now(); }
  void mark(const char *what);
  void report(const char *head);
};
extern thread_local CallClock *tl_call_clock;
inline void clock_mark(const char *what) { if (tl_call_clock && tl_call_clock->on) tl_call_clock->mark(what); }

// ---- kernel launchers (mxa_kernels.hip)
// recode raw PLINK rows (src pitch arbitrary) into the padded z-coded device layout
int launch_recode(const uint8_t *d_src, size_t src_pitch, long row0, long nrows, long k, long k_bit_offset,
                  PackedMatrix &dst, hipStream_t s);
// denormal-operand mode of k_gemm (MODE 2): B columns are scaled to just below 2^kDenUp, the genotype operand is z * 2^-1074
constexpr int kDenUp = 900;
// A product z * b' * 2^-1074 (b' = the scaled entry) is a NORMAL double, i.e. exact, while |b'| >= 2^52: entries up to 847 binades below
// their column's largest one.  Columns whose non-zero entries span more than kDenMaxSpan binades (or hold inf / NaN) are multiplied with
// the two-instruction conversion (MODE 0, plain fp64 operands, no scaling) instead: launch_colexp raises a device flag, the MODE 0 chain
// is enqueued behind the MODE 2 one with run_if_set = that flag and overwrites its partial sums; k_finish skips the scale-back.
constexpr int kDenMaxSpan = 800;
// E[j] = binary exponent of the largest |entry| of column j (frexp convention) + bias; d_part: 64 * n doubles of scratch
// d_flag (nullable, device int): range guard of the exact int8 slicing, see k_colexp_final; then d_part needs 128 * n doubles
constexpr int kI8ExactMaxDigits = 24;   // engine 4: more digits than this cost more than the fp64 MFMA path
int launch_colspan(const double *dB, long ldb, long k, int n, double *d_part, int *d_out3, hipStream_t s);   // engine 4: span / e_max / non-finite verdict for the host
int launch_colexp_from_part(const double *d_part, int n, int *d_E, int bias, hipStream_t s);
int launch_colexp(const double *dB, long ldb, long k, int n, double *d_part, int *d_E, int bias, hipStream_t s, int *d_flag = nullptr, int max_span = 0,
                  int min_emax = 0, bool reset_flag = true);
// d_E (nullable): per-column exponents for the denormal-operand mode
// K-steps (16 genotypes) [S0, S0 + S_cnt) only; S_cnt < 0: to the end
// run_if_set (nullable, device int): the kernel does nothing unless *run_if_set != 0
int launch_pack_B(const double *dB, long ldb, long k, int n, double *dBp, long k_pad, int n_pad, int c, hipStream_t s, const int *d_E = nullptr, long S0 = 0,
                  long S_cnt = -1, const int *run_if_set = nullptr, bool rowscale = false);
int launch_colsums(const double *dB, long ldb, long k, int n, const double *d_f /*nullable*/, double *d_part,
                   double *d_sumB, double *d_sumfB, hipStream_t s);
// ksplits_like: take the K pieces of another plan (row ranges of one product: identical sums)
GemmPlan plan_gemm(long m, long k_pad, int n, const GemmPlan *ksplits_like = nullptr);
// conversion variant of k_gemm for a tile of c column groups: 2 (v_bfe_u32) or 3 (v_and_b32 + B rows pre-scaled); MXA_GEMM_MODE overrides
int gemm_default_mode(int c);
// K splits [split_begin, split_end) only (split_end < 0: all)
// run_if_set (nullable, device int; MODE 0 only): the kernel does nothing unless *run_if_set != 0 (fallback of the denormal-operand mode)
// d_ctr: 9 ints of device memory for the work queues of this launch (zeroed here, on s); must not be shared with a launch that may run at the same time
// tr: transposed operand (k_gemm<..., TR>): the output rows are G's COLUMNS (p planned for m = G.k) and K runs over G's rows -- the 'N' product from the
// SNP-major copy; modes 0 and 2 only
// p_split0: the K split whose partial sums land at dP (grouped K splits keep one GROUP of splits in the buffer: p_split0 = the group's first split;
// the host-operand pipeline addresses the whole buffer: 0)
int launch_gemm(const PackedMatrix &G, const double *dBp, double *dP, const GemmPlan &p, int mode, hipStream_t s, int *d_ctr, int split_begin = 0, int split_end = -1,
                const int *run_if_set = nullptr, bool tr = false, int p_split0 = 0);
GemmPlan plan_lut(long m, long k_pad, int n);
// run_if_set (nullable, device int): the kernel does nothing unless *run_if_set != 0 (fallback of the guarded small-n route)
int launch_lut(const PackedMatrix &G, const double *dB, long ldb, int n, double *dP, const GemmPlan &p, hipStream_t s, const int *run_if_set = nullptr);
// fill_rows: rows [m, fill_rows) of every column are zero-filled (fill_rows <= ldc)
int launch_finish(const double *dP, const GemmPlan &p, long m, int n, double *dC, long ldc, long fill_rows, int mode_trans,
                  bool centered, const double *d_sumB, const double *d_sumfB, const double *d_f, hipStream_t s, const int *d_E = nullptr, int e_splits = 0,
                  int e_stride = 0, const int *run_if_set = nullptr, const int *unscaled_if_set = nullptr, int group_splits = 0, int group = 0);
// group_splits > 0: dP holds only that many splits (one GROUP of the plan's K splits); group bit 0: continue the running sum kept in C, bit 1: not the
// last group -- the raw sum is stored (no scale-back, no centring).  See gemm_grouped (mxa_api.cpp).
// unscaled_if_set (nullable, device int): when *unscaled_if_set != 0 the partial sums were produced WITHOUT the operand scaling (MODE 0 fallback): d_E is ignored
// per-device one-time hipFuncSetAttribute(MaxDynamicSharedMemorySize): function attributes are per device, `mask` has one bit per device
int ensure_dyn_lds(const void *func, int bytes, unsigned long long *mask);
int launch_transpose_2bit(const uint8_t *d_in, long rows, long cols, uint8_t *d_out, hipStream_t s);
int launch_allele_freq(const uint8_t *d_plink, long snps, long indiv, double *d_f, hipStream_t s);
// crossproduct
int launch_plink_lut(uint8_t *d, size_t nbytes, hipStream_t s);
struct XPost;   // element-wise map of the crossproduct epilogue (GRM / LD post-processing, mxa_crossprod.hip)
int crossprod_device(const uint8_t *d_X, long k, long rows, size_t pitch, double *d_ans, hipStream_t s, long c_begin, long c_end, bool upper_only, long ld, bool f4,
                     int post_kind = 0, const XPost *post = nullptr);
// sparse_times_plink (mxa_sparse.hip): one slab of C columns [e_base, e_base + e_count) (e_base a multiple of 4), C slab pointer = column e_base
int launch_sparse_times_plink(const uint8_t *dP, size_t pitch, long entries, int nIdx, const int *d_rowIdx, const int *d_colIdx, const double *d_val,
                              double *dC_slab, long ldc, long e_base, long e_count, hipStream_t s);
// opt-in engine (mxa_set_engine(1) / MXA_ENGINE=i8, mxa_gemm_i8.hip): whole product by exact int8 slicing of B.
// Asynchronous on s; ev0/ev1 (optional) are recorded around the dominant kernel.
// guard: check on the device that every column of B is represented EXACTLY by the slicing (finite, dynamic range inside the digits, no
// underflow in the recombination).  guard = 1: the host reads the verdict (one 4-byte copy + sync); if it is negative nothing is computed
// and 2 is returned (the caller takes the fp64 path).  guard = 2: no host round trip -- the three kernels of the chain test the device
// flag themselves and do nothing when it is set; 3 is returned with *flag_out = the device flag, and the caller enqueues the fp64
// fallback with run_if_set = that flag.
// S_override > 0 (engine 4): that many digits per column for every n (the caller has already proved them sufficient: guard = 0).
// colsum_scratch (guard = 2, n <= 2 only; 128 n doubles): the column sums of the centring term are computed HERE, in the same pass over B as
// the exponents and the guard (k_colstats_partial; finished inside k_slice_B) -- the caller must not have launched launch_colsums for them
// chain (guard = 2, round 5): the verdict is a CLASS decided on the device (mxa_gemm_i8.hip: SliceFused) -- class 0 / 1: B is represented exactly by S0 / S1
// digits per column (S1 = 0: no second class), class 2: neither.  One call enqueues the chain of ONE class (my_class; its kernels do nothing unless the
// verdict is that class); `first`: this call also launches the statistics pass, publishes E, the column sums and the three flag words, and carries the fp64
// chains of class 2 inside its k_slice_B launch.  The caller enqueues the chains of all classes back to back; nothing waits for the host.
struct I8Chain { int S0 = 0, S1 = 0, my_class = 0; bool first = true; bool fp64_rows = true; };   // fp64_rows: verdict class 2 is served by the fp64 chains inside the first chain's k_slice_B launch (false: by gated launches of the caller)
int gemm_i8_device(const PackedMatrix &G, bool trans, int n, const double *dB, long ldb, double *dC, long ldc, long fill_rows, bool centered, double *d_sumB,
                   double *d_sumfB, const double *d_f, Workspace &w, hipStream_t s, hipEvent_t ev0, hipEvent_t ev1, int *splits_out, int guard = 0,
                   const int **flag_out = nullptr, double *colsum_scratch = nullptr, int S_override = 0, const PackedMatrix *G_tn = nullptr,
                   double *stats_part = nullptr, const I8Chain *chain = nullptr);
// workspace for a chain with S digits (0: the default of n), before anything is enqueued; 2: the transposed-operand form declines this tile count
int gemm_i8_reserve(const PackedMatrix &G, int n, int S, const PackedMatrix *G_tn, Workspace &w, hipStream_t s);
// stats_part (guard = 0): the column maxima / minima of B are already there (k_colmax_partial's layout, 128 n doubles; the caller's launch_colspan made them to
// choose S): only the exponents are derived, no second pass over B
// G_tn: the OTHER stored orientation (rows = the K index); the main kernel then runs in the transposed-operand form k_gemm_i8_tn, one launch per tile of 32
// expanded columns.  Returns 2 (nothing enqueued) when that would take more passes than the fp64 MFMA tile costs or the plan has several column chunks.

// mxa_dense.hip: dense fp64 MFMA building blocks of the solver twin
int launch_dgemm(bool ta, bool tb, long M, long N, long K, double alpha, const double *A, long lda, const double *B, long ldb, double beta, double *C, long ldc,
                 bool lower_only, hipStream_t s);
int launch_potrf_inv_block(double *A, long ld, int nb, long offset, int *info, double *inv_out, hipStream_t s);

}  // namespace mxa
