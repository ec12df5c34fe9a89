// mxa_api.cpp -- host side of the C ABI (include/miraculix_amd.h): options singleton, PLINK .bed staging,
// per-call orchestration.  Mirrors the reference's L3/L2 layers (src/miraculix/5codesAPI.c,
// src/miraculix/5codesChar.cc:165-449) and the host part of src/cuda/dgemm_compressed_cuda.cu, re-designed so that
// nothing is allocated, created or destroyed per multiply and operands may already live in HBM.
//
// No CPU fallback: every compute entry needs a HIP device and fails loudly without one.
#include "../../include/miraculix_amd.h"
#include "mxa_internal.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

namespace mxa {

// ------------------------------------------------------------------------------------------------ state
static int g_err = 0;
static char g_errmsg[512] = "";

Options &options() { static Options o; return o; }
Profile &profile() { static Profile p; return p; }
Geometry &last_geometry() { static Geometry g; return g; }
static bool g_profile_on = true;
// multiply engine (include/miraculix_amd.h, mxa_set_engine): 0 = fp64 MFMA (default), 1 = int8 slicing, 3 = fp64 only, 4 = exact int8 slicing; ids 2 and 5 (rounds 3-5) are retired
static std::atomic<int> g_engine{[] {
  const char *e = getenv("MXA_ENGINE");
  if (e && std::string(e) == "i8") return 1;
  if (e && std::string(e) == "f64-strict") return 3;
  if (e && std::string(e) == "i8-exact") return 4;
  return 0;
}()};

int env_print_level() {  // reference: cuda_utils.cu:44-52, env PRINT_LEVEL
  const char *e = getenv("PRINT_LEVEL");
  return e ? atoi(e) : 0;
}

// Banner of the reference (cuda_utils.cu:64-81: name, compile date, git commit), once per process; printed under PRINT_LEVEL > 0 /
// print_details only -- the reference prints it by default, this library is quiet unless asked.
#ifndef MXA_COMMIT_ID
#define MXA_COMMIT_ID "unknown"
#endif
void print_compile_info(const char *what) {
  static std::atomic<bool> done{false};
  if (done.exchange(true)) return;
  printf("------------------------------------------------------------\n\tmiraculix_amd (MI355X / gfx950) - %s\nCompiled on %s %s, git commit %s\n"
         "------------------------------------------------------------\n", what, __DATE__, __TIME__, MXA_COMMIT_ID);
}

// The status is per process, like the reference's (its entries return void and print).  Worker threads of a multi-device object
// report through the same state: the mutex keeps the message intact, the first error of a call wins.
static std::mutex g_prof_mutex;   // profile() / last_geometry() are written by the worker threads of multi-device objects too
static std::mutex g_err_mutex;
void clear_error() { std::lock_guard<std::mutex> lk(g_err_mutex); g_err = 0; g_errmsg[0] = 0; }

void set_error(int code, const char *fmt, ...) {
  char msg[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(msg, sizeof(msg), fmt, ap);
  va_end(ap);
  fprintf(stderr, "miraculix_amd: %s\n", msg);
  std::lock_guard<std::mutex> lk(g_err_mutex);
  if (g_err == 0) { g_err = code; memcpy(g_errmsg, msg, sizeof(g_errmsg)); }
}

bool check_hip(hipError_t e, const char *func, int line) {
  if (e == hipSuccess) return true;
  // wording follows the reference's checkError (cuda_utils.cu:83-90)
  set_error(100 + (int)e, "Internal error in function %s at line %d: %s", func, line, hipGetErrorString(e));
  return false;
}

void debug_info(const char *fmt, ...) {
  if (env_print_level() <= 0 && options().print_level <= 0) return;
  va_list ap;
  va_start(ap, fmt);
  printf("\t ");
  vprintf(fmt, ap);
  printf("\n");
  va_end(ap);
}

thread_local CallClock *tl_call_clock = nullptr;
double CallClock::now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
void CallClock::mark(const char *what) {
  if (!on) return;
  const double t = now();
  if (len < (int)sizeof(line) - 48) len += snprintf(line + len, sizeof(line) - len, " %s %.3f", what, (t - last) * 1e3);
  last = t;
}
void CallClock::report(const char *head) {
  if (!on) return;
  printf("\t %s: total %.3f ms |%s (ms)\n", head, (now() - t0) * 1e3, line);
  on = false;
}

static Handle *as_handle(void *p, const char *who) {
  Handle *h = reinterpret_cast<Handle *>(p);
  if (!h || h->magic != kMagic) {
    set_error(2, "%s: invalid or uninitialised compressed object", who);
    return nullptr;
  }
  return h;
}

int ptr_location(const void *p, int *dev) {
  if (dev) *dev = -1;
  if (!p) return 0;
  hipPointerAttribute_t attr;
  hipError_t e = hipPointerGetAttributes(&attr, p);
  if (e != hipSuccess) { (void)hipGetLastError(); return 0; }
  if (attr.type == hipMemoryTypeDevice || attr.type == hipMemoryTypeManaged) { if (dev) *dev = attr.device; return 1; }
  return 0;
}
static bool is_device_ptr(const void *p) { return ptr_location(p, nullptr) == 1; }

// device selection: env CUDA_DEVICE is honoured like the reference (cuda_utils.cu:187-247) but visibility variables
// need not be set; HIP_DEVICE takes precedence.
int select_device() {
  int count = 0;
  hipError_t e = hipGetDeviceCount(&count);
  if (e != hipSuccess || count <= 0) {
    (void)hipGetLastError();
    set_error(10, "no HIP device available (hipGetDeviceCount: %s). This engine is GPU-only.", hipGetErrorString(e));
    return -1;
  }
  int cur = 0;
  (void)hipGetDevice(&cur);
  int dev = cur;
  const char *d = getenv("HIP_DEVICE");
  if (!d) d = getenv("CUDA_DEVICE");
  if (d) dev = atoi(d);
  if (dev < 0 || dev >= count) {
    set_error(11, "The requested device %d is not visible to the HIP runtime (%d devices).", dev, count);
    return -1;
  }
  if (!check_hip(hipSetDevice(dev), __func__, __LINE__)) return -1;
  return dev;
}

static int grow(double **p, size_t *cap, size_t need_elems) {
  if (*cap >= need_elems) return 0;
  if (*p) { MXA_HIP(hipFree(*p)); *p = nullptr; *cap = 0; }
  MXA_HIP(hipMalloc(reinterpret_cast<void **>(p), need_elems * sizeof(double)));
  *cap = need_elems;
  return 0;
}

// ------------------------------------------------------------------------------------------------ staging
// Allocate the padded device layout for `rows` x `k` genotypes and fill it from raw PLINK rows (host or device).
static int stage_matrix(PackedMatrix &M, const uint8_t *src, size_t src_pitch, long rows, long k, hipStream_t s) {
  M.rows = rows; M.k = k;
  M.rows_pad = (rows + kRowAlign - 1) / kRowAlign * kRowAlign;
  M.k_pad = (k + kSlabK - 1) / kSlabK * kSlabK;
  M.pitch = (size_t)M.k_pad / 4;
  M.nslabs = M.k_pad / kSlabK;
  const size_t bytes = (size_t)M.rows_pad * M.pitch;
  MXA_HIP(hipMalloc(reinterpret_cast<void **>(&M.d), bytes));
  MXA_HIP(hipMemsetAsync(M.d, 0, bytes, s));
  const long row_bytes = (k + 3) / 4;
  int src_dev = -1;
  if (ptr_location(src, &src_dev) == 1) {
    int cur = 0;
    MXA_HIP(hipGetDevice(&cur));
    if (src_dev != cur) {   // the recode kernel reads the other GPU's memory over xGMI
      if (!enable_peer(cur, src_dev)) { set_error(15, "plink2compressed: device %d cannot read the source matrix in the memory of device %d (peer access not available or could not be enabled)", cur, src_dev); return 1; }
      if (sync_foreign_producer(src_dev)) return 1;   // the matrix may still be being written by work on the source device's default stream
    }
    if (launch_recode(src, src_pitch, 0, rows, k, 0, M, s)) return 1;
    MXA_HIP(hipStreamSynchronize(s));
    return 0;
  }
  // host source: stream row chunks through a device bounce buffer (compacting the pitch), recode on device
  const size_t chunk_bytes = (size_t)256 << 20;
  long chunk_rows = std::max<long>(1, (long)(chunk_bytes / (size_t)std::max<long>(1, row_bytes)));
  chunk_rows = std::min(chunk_rows, rows);
  uint8_t *bounce = nullptr;
  MXA_HIP(hipMalloc(reinterpret_cast<void **>(&bounce), (size_t)chunk_rows * row_bytes));
  int rc = 0;
  for (long r0 = 0; r0 < rows && !rc; r0 += chunk_rows) {
    const long nr = std::min(chunk_rows, rows - r0);
    hipError_t e;
    if ((size_t)row_bytes == src_pitch) e = hipMemcpyAsync(bounce, src + (size_t)r0 * src_pitch, (size_t)nr * row_bytes, hipMemcpyHostToDevice, s);
    else e = hipMemcpy2DAsync(bounce, row_bytes, src + (size_t)r0 * src_pitch, src_pitch, row_bytes, nr, hipMemcpyHostToDevice, s);
    if (!check_hip(e, __func__, __LINE__)) { rc = 1; break; }
    rc = launch_recode(bounce, row_bytes, r0, nr, k, 0, M, s);
    if (!rc && !check_hip(hipStreamSynchronize(s), __func__, __LINE__)) rc = 1;
  }
  (void)hipFree(bounce);
  return rc;
}

// dimensions of a packed matrix that is NOT stored (single-orientation objects): what stage_matrix would have set, d == nullptr
static void describe_matrix(PackedMatrix &M, long rows, long k) {
  M.d = nullptr; M.rows = rows; M.k = k;
  M.rows_pad = (rows + kRowAlign - 1) / kRowAlign * kRowAlign;
  M.k_pad = (k + kSlabK - 1) / kSlabK * kSlabK;
  M.pitch = (size_t)M.k_pad / 4;
  M.nslabs = M.k_pad / kSlabK;
}

// Both orientations from the SNP-major PLINK matrix alone: the raw bytes are brought to the device once (a host source: one compact upload;
// a device source is used in place), recoded into the SNP-major object, transposed on the device (raw PLINK codes, so a missing 01 stays a
// missing 01) and the transposed bytes recoded into the individual-major object -- bit-identical to staging a caller-supplied transposed
// copy.  Temporary device memory: the raw matrix (host sources only) + its transpose, released before the workspace is allocated.
static int stage_from_snp_major(Handle *h, const uint8_t *plink, size_t plink_pitch) {
  const long snps = h->snps, indiv = h->indiv;
  const size_t bps = ((size_t)indiv + 3) / 4, bpi = ((size_t)snps + 3) / 4;
  hipStream_t s = h->stream;
  int src_dev = -1;
  const bool src_on_device = ptr_location(plink, &src_dev) == 1;
  uint8_t *d_raw = nullptr, *d_raw_t = nullptr;
  int rc = 0;
  const uint8_t *d_src = plink;
  size_t d_pitch = plink_pitch;
  if (!src_on_device || src_dev != h->device || plink_pitch != bps) {   // compact copy on this device (the transpose kernels want pitch = ceil(indiv/4))
    if (src_on_device && src_dev != h->device && sync_foreign_producer(src_dev)) return 1;
    MXA_HIP(hipMalloc(reinterpret_cast<void **>(&d_raw), (size_t)snps * bps));
    hipError_t e = plink_pitch == bps ? hipMemcpyAsync(d_raw, plink, (size_t)snps * bps, hipMemcpyDefault, s)
                                      : hipMemcpy2DAsync(d_raw, bps, plink, plink_pitch, bps, snps, hipMemcpyDefault, s);
    if (!check_hip(e, __func__, __LINE__) || !check_hip(hipStreamSynchronize(s), __func__, __LINE__)) { (void)hipFree(d_raw); return 1; }
    d_src = d_raw; d_pitch = bps;
  }
  rc = stage_matrix(h->snp_major, d_src, d_pitch, snps, indiv, s);
  if (!rc && !check_hip(hipMalloc(reinterpret_cast<void **>(&d_raw_t), (size_t)indiv * bpi), __func__, __LINE__)) rc = 1;
  if (!rc) rc = launch_transpose_2bit(d_src, snps, indiv, d_raw_t, s);
  if (!rc && !check_hip(hipStreamSynchronize(s), __func__, __LINE__)) rc = 1;
  if (d_raw) { (void)hipFree(d_raw); d_raw = nullptr; }
  if (!rc) rc = stage_matrix(h->ind_major, d_raw_t, bpi, indiv, snps, s);
  if (d_raw_t) (void)hipFree(d_raw_t);
  return rc;
}

void destroy_handle(Handle *h) {
  if (!h) return;
  {   // mxa_last_path() must not read a flag of this object once its memory is gone
    std::lock_guard<std::mutex> lk(g_prof_mutex);
    Geometry &geo = last_geometry();
    if (geo.d_flag && h->ws.d_denflag && geo.d_flag >= h->ws.d_denflag && geo.d_flag < h->ws.d_denflag + 4) { geo.d_flag = nullptr; if (geo.path == 4) geo.path = 2; }
  }
  (void)hipSetDevice(h->device);
  if (h->stream) (void)hipStreamSynchronize(h->stream);
  void *ptrs[] = {h->snp_major.d, h->ind_major.d, h->d_f, h->ws.d_Bstage, h->ws.d_Cstage, h->ws.d_Bp, h->ws.d_P, h->ws.d_colpart, h->ws.d_i8, h->ws.d_tmp, h->ws.d_exp, h->ws.d_denflag};
  for (void *p : ptrs) if (p) (void)hipFree(p);
  for (hipEvent_t e : {h->ev0[0], h->ev0[1], h->ev1[0], h->ev1[1], h->ev_in[0], h->ev_in[1], h->ev_out[0], h->ev_out[1]}) if (e) (void)hipEventDestroy(e);
  for (hipEvent_t e : h->pev) if (e) (void)hipEventDestroy(e);
  for (hipStream_t ps : h->pipe) if (ps) (void)hipStreamDestroy(ps);
  if (h->stream) (void)hipStreamDestroy(h->stream);
  free(h->h_f);
  h->magic = 0;
  delete h;
}

// The partial-sum workspace is sized by ensure_workspace for the plans of n columns; a product that peels odd columns multiplies fewer columns
// with a DIFFERENT plan (tile width, K pieces), which can need more room than the unpeeled one (100 000 x 30 000, n = 9..11: 12.81 M doubles
// against 11.96 M).  Every launch path calls this with the plan it is about to launch.
static int ensure_partials(Workspace &w, const GemmPlan &p, hipStream_t s) {
  const size_t need = (size_t)p.splits * p.n_pad * p.m_pad;
  if (need <= w.cap_P) return 0;
  MXA_HIP(hipStreamSynchronize(s));   // earlier products on this stream may still be reading the old buffer
  return grow(&w.d_P, &w.cap_P, need);
}

// How many doubles of split-K partial sums a product may hold (round 5).  A plan cuts K into pieces whose B slabs stay in an XCD's L2 (plan_gemm), and
// every piece leaves a partial result of the whole output: at BASELINE config 4's full extent (5M SNPs x 200k individuals, n = 128) that is 18 x 5.1 GB
// for 'T' and 407 x 0.2 GB for 'N' beside 250 GB of packed genotypes.  Products whose partials exceed this budget run their K splits in GROUPS
// (gemm_device, "GROUPED K splits"): same pieces, same order of additions, the running sum kept in C.  Budget: everything up to 4 GiB; beyond that at
// most 16 GiB (a group costs one more read + write of C: 18 groups of config 4's 'T' 1.3 % of the call, 6 groups 0.4 %) and at most what the device has
// free right now (the old buffer counted as free: it is released before the new one is allocated) less 2 GiB; never less than ONE split's partials.
// MXA_P_BUDGET_MB overrides (tests force the grouped path on small products).
// The memory-dependent part is decided ONCE per object (Workspace::big_budget, at the first product that needs it): later products reuse it, so a solver
// loop never re-grows the buffer because more memory happens to be free (hipFree + device sync + a multi-GiB hipMalloc in the middle of the loop).
static size_t partial_budget(Workspace &w, size_t need, size_t one_split) {
  const char *e_mb = getenv("MXA_P_BUDGET_MB");   // test knob, read per product: tests/test_grouped_and_incremental_gpu.py switches it on one object
  const long env_mb = e_mb ? atol(e_mb) : -1L;
  if (env_mb >= 0) return std::max(one_split, std::min(need, (size_t)env_mb * (1u << 20) / sizeof(double)));
  if (need * sizeof(double) <= ((size_t)4 << 30) || need <= w.cap_P) return need;
  if (!w.big_budget) {
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { (void)hipGetLastError(); return need; }
    const size_t avail = free_b + w.cap_P * sizeof(double), margin = (size_t)2 << 30, soft = (size_t)16 << 30;
    w.big_budget = std::max<size_t>(1, std::min(soft, avail > margin ? avail - margin : 0) / sizeof(double));
  }
  return std::max(one_split, std::min(need, std::max(w.big_budget, (size_t)w.cap_P)));
}

static int ensure_workspace(Handle *h, int n) {
  // sized for the larger of the two products, like the reference's size_buffer (dgemm_compressed_cuda.cu:77)
  Workspace &w = h->ws;
  const long kmax_pad = std::max(h->snp_major.k_pad, h->ind_major.k_pad);
  GemmPlan pn = plan_gemm(h->indiv, h->ind_major.k_pad, n), pt = plan_gemm(h->snps, h->snp_major.k_pad, n);
  const size_t bp = (size_t)kmax_pad * std::max(pn.n_pad, pt.n_pad);
  GemmPlan ln = plan_lut(h->indiv, h->ind_major.k_pad, std::min(n, 4)), lt = plan_lut(h->snps, h->snp_major.k_pad, std::min(n, 4));
  size_t pp = std::max(std::max((size_t)pn.splits * pn.n_pad * pn.m_pad, (size_t)pt.splits * pt.n_pad * pt.m_pad),
                       std::max((size_t)ln.splits * ln.n_pad * ln.m_pad, (size_t)lt.splits * lt.n_pad * lt.m_pad));
  if (grow(&w.d_Bp, &w.cap_Bp, bp)) return 1;
  // very large products keep only a GROUP of their K splits' partial sums at a time (partial_budget, gemm_grouped)
  pp = partial_budget(w, pp, std::max(std::max((size_t)pn.n_pad * pn.m_pad, (size_t)pt.n_pad * pt.m_pad), std::max((size_t)ln.splits * ln.n_pad * ln.m_pad, (size_t)lt.splits * lt.n_pad * lt.m_pad)));
  if (grow(&w.d_P, &w.cap_P, pp)) return 1;
  if (grow(&w.d_colpart, &w.cap_colpart, (size_t)n * (64 * 2 + 2) + 16)) return 1;
  if (!w.d_denflag) {   // flags + the work-queue counters of k_gemm, one small block for the life of the handle
    MXA_HIP(hipMalloc(reinterpret_cast<void **>(&w.d_denflag), (16 + 16 * 16) * sizeof(int)));
    MXA_HIP(hipMemset(w.d_denflag, 0, (16 + 16 * 16) * sizeof(int)));
    w.d_ctr = w.d_denflag + 16;
  }
  return 0;
}

// digits per column of the guarded exact int8 route by column count (gemm_device: guarded_small): class 0 / class 1
static const int kSmallS0[7] = {0, 32, 16, 10, 16, 12, 10}, kSmallS1[7] = {0, 0, 0, 21, 24, 19, 16};

// The int8 workspace the guarded route of an n-column product will ask for (n <= 6: the whole product; n = 4q + r: the r peeled columns), reserved when
// the object is made instead of inside the first product (1.1-1.4 ms of hipMalloc in the first dgemm_compressed call of the reference's harness).
static int reserve_small_routes(Handle *h, int n) {
  const int nc = n <= 6 ? n : (n & 3);
  if (nc <= 0 || g_engine.load() != 0) return 0;
  for (int trans = 0; trans < 2; trans++) {
    const PackedMatrix &G = trans ? h->snp_major : h->ind_major;
    if (G.k < 128) continue;
    const PackedMatrix *G_tn = (h->single && !trans) ? &h->snp_major : nullptr;
    if (gemm_i8_reserve(G, nc, kSmallS0[nc], G_tn, h->ws, h->stream) == 1) return 1;
    if (kSmallS1[nc] && gemm_i8_reserve(G, nc, kSmallS1[nc], G_tn, h->ws, h->stream) == 1) return 1;
  }
  return 0;
}

thread_local int tl_single_override = -1;
thread_local bool tl_no_warmup = false;   // one-shot objects (dgemm_plink) skip the warm-up products
static thread_local bool tl_in_warmup = false;

// First-call costs belong to plink2compressed, not to the first dgemm_compressed (round 6).  The phase clock of the reference's harness showed its FIRST product
// through the plain ABI at 13.5-15.2 ms where the following ones take 4.0 (250k x 50k x 10): 7.2-9.3 ms inside the first device-to-host copy of C on the
// object's stream (the runtime's staged pageable copy; independent of the size, not page faults), 0.8-1.5 ms in the first upload of B, 1.2-1.4 + 0.55 ms of
// first launches of the kernels on the path (profiles/r06_harness_phase_clock.txt).  One 'N' and one 'T' product with max_n columns of zeros, operands where
// the caller's matrices were (host scratch for a host caller: exactly the path its calls will take), pays all of that here -- bounded: the column count
// is cut until a product is estimated below 12.5 ms, and objects whose single-column product is slower than that (HBM-bound at > 60 GB) are not warmed.
// MXA_WARMUP=0 turns it off.  Nothing is timed or counted (timing = false).
static int gemm_device(Handle *h, bool trans, int n, const double *dB, long ldb, double *dC, long ldc, long fill_rows, hipStream_t s, bool timing);
static int warm_up(Handle *h, bool host_caller) {
  const char *e = getenv("MXA_WARMUP");
  if (tl_no_warmup || (e && atoi(e) == 0)) return 0;
  if (options().centered && !h->has_f) return 0;
  const double cells = (double)h->snps * (double)h->indiv;
  auto est = [&](int nn) { return std::max(2.0 * cells * nn / 70e12, cells / 4.0 / 5e12); };
  int n = h->max_n;
  while (n > 1 && est(n) > 0.0125) n = n > 8 ? (n / 2 + 3) / 4 * 4 : n - 1;
  if (est(n) > 0.0125) return 0;
  const size_t cnt = (size_t)std::max(h->snps, h->indiv) * n;
  int rc = 0;
  tl_in_warmup = true;
  if (host_caller) {
    double *hb = (double *)calloc(cnt, sizeof(double)), *hc = (double *)malloc(cnt * sizeof(double));
    if (hb && hc) {
      rc = gemm_any(h, false, n, hb, h->snps, hc, h->indiv, h->indiv, true, false);
      if (!rc) rc = gemm_any(h, true, n, hb, h->indiv, hc, h->snps, h->snps, true, false);
    }
    free(hb); free(hc);
  } else {
    Workspace &w = h->ws;
    if (grow(&w.d_Bstage, &w.cap_Bstage, cnt) || grow(&w.d_Cstage, &w.cap_Cstage, cnt)) { tl_in_warmup = false; return 1; }
    if (!check_hip(hipMemsetAsync(w.d_Bstage, 0, cnt * sizeof(double), h->stream), __func__, __LINE__)) { tl_in_warmup = false; return 1; }
    rc = gemm_device(h, false, n, w.d_Bstage, h->snps, w.d_Cstage, h->indiv, h->indiv, h->stream, false);
    if (!rc) rc = gemm_device(h, true, n, w.d_Bstage, h->indiv, w.d_Cstage, h->snps, h->snps, h->stream, false);
    if (!rc && !check_hip(hipStreamSynchronize(h->stream), __func__, __LINE__)) rc = 1;
  }
  { std::lock_guard<std::mutex> lk(g_prof_mutex); last_geometry() = Geometry(); }   // mxa_last_path / mxa_last_geometry describe the CALLER'S products
  harvest_profile(h);
  h->prof = ObjectProfile();            // ... and so do the per-object counters
  tl_in_warmup = false;
  return rc;
}

int single_orientation_policy() {
  // Round 5: ONE packed copy (SNP-major) is the default -- both products read it at the rate two copies reach (fp64 MFMA: plain form = transposed form =
  // 0.957-0.962 of the peak; the CG step within 2 %), for half the HBM and half the staging.  MXA_SINGLE_ORIENTATION=0 asks for both copies (the opt-in: 'N'
  // with 4 <= n <= 6 columns then runs the plain int8 kernel in one pass instead of two); they are kept when they fit the device's free memory, otherwise
  // one copy is kept and a line on stderr says so (the reference reports "Not enough device memory" there, cuda_utils.cu:162-185).
  const char *e = getenv("MXA_SINGLE_ORIENTATION");
  if (!e || !*e) return 1;
  return atoi(e) != 0 ? 1 : 2;
}

size_t object_footprint(long snps, long indiv, int max_n, bool single) {
  return (size_t)((snps + kRowAlign) * ((indiv + kSlabK) / 4)) + (single ? 0 : (size_t)((indiv + kRowAlign) * ((snps + kSlabK) / 4))) +
         (size_t)3 * sizeof(double) * (size_t)std::max(snps, indiv) * (size_t)std::max(max_n, 1);
}

int create_handle(const uint8_t *plink, size_t plink_pitch, const uint8_t *plink_t, size_t plink_t_pitch, long snps, long indiv, const double *f,
                  int max_n, void **out, int device) {
  if (out) *out = nullptr;
  if (!out) { set_error(1, "plink2compressed: compressed is NULL"); return 1; }
  if (!plink) { set_error(1, "plink2compressed: plink is NULL"); return 1; }
  // plink_transposed NULL or the same pointer as plink: the reference's CPU path never reads it and its Fortran benchmark passes the same
  // pointer twice (5codesChar.cc:368-393, utils/benchmark/benchmark.f90:185).  Only the SNP-major matrix crosses PCIe then; the
  // individual-major copy is produced on the device (k_transpose_2bit*), as mxa_bed2compressed does.
  const bool one_pointer = !plink_t || plink_t == plink;
  if (snps <= 0 || indiv <= 0) { set_error(1, "plink2compressed: snps and indiv must be positive"); return 1; }
  Options &o = options();
  if (!o.set) {  // reference default before any user call: gpu when compiled with CUDA, centred (5codesChar.cc:127-143)
    o.gpu = true; o.centered = true; o.set = true;
  }
  if (!o.gpu) { set_error(14, "plink2compressed: setOptions_compressed was called with use_gpu=0; this library has no CPU engine"); return 1; }
  int dev = device;
  if (dev < 0) dev = select_device();
  else if (!check_hip(hipSetDevice(dev), __func__, __LINE__)) dev = -1;
  if (dev < 0) return 1;
  if (env_print_level() > 0 || o.print_level > 0) {
    hipDeviceProp_t prop;
    print_compile_info("dgemm_compressed");
    if (hipGetDeviceProperties(&prop, dev) == hipSuccess) printf("miraculix_amd - dgemm_compressed: using device %s [%s] (device no %d).\n", prop.name, prop.gcnArchName, dev);
  }
  // memory pre-flight like checkDevMemory (cuda_utils.cu:162-185).  Where the reference gives up -- two packed copies do not fit -- this build keeps
  // the SNP-major copy alone if THAT fits (MXA_SINGLE_ORIENTATION unset / auto; Handle::single): both products then read the one copy.
  size_t free_b = 0, total_b = 0;
  MXA_HIP(hipMemGetInfo(&free_b, &total_b));
  const int policy = tl_single_override >= 0 ? tl_single_override : single_orientation_policy();
  const size_t need_two = object_footprint(snps, indiv, max_n, false), need_one = object_footprint(snps, indiv, max_n, true);
  // what the staging itself holds on top at its peak: a host source goes through a 256 MB bounce buffer; the one-pointer shape keeps the raw transposed
  // block (about one packed copy) until the second copy is built
  int src_dev0 = -1;
  const size_t bounce = ptr_location(plink, &src_dev0) == 1 ? 0 : (size_t)256 << 20;
  const size_t tmp_two = one_pointer ? (size_t)indiv * (((size_t)snps + 3) / 4) + bounce : bounce, tmp_one = bounce;
  const bool single0 = policy == 1 || (policy == 2 && need_two + tmp_two > free_b && need_one + tmp_one <= free_b);
  if (single0 && policy == 2)   // always said: the caller asked for two copies (mxa_single_orientation() reports what the object holds)
    fprintf(stderr, "miraculix_amd - dgemm_compressed: two packed copies need %.1f GB, %.1f GB are free: keeping the SNP-major copy only (%.1f GB).\n", need_two / 1e9, free_b / 1e9, need_one / 1e9);
  const size_t need = single0 ? need_one : need_two;
  if (need > free_b) {
    set_error(12, "Not enough device memory available. Required %zu GB, free %zu GB, total on device %zu GB", need >> 30, free_b >> 30, total_b >> 30);
    return 1;
  }
  Handle *h = new Handle();
  h->device = dev; h->snps = snps; h->indiv = indiv; h->max_n = std::max(max_n, 1);
  // a BLOCKING stream: it orders itself against the legacy default stream, so device-resident B produced by work the caller
  // enqueued on the default stream (e.g. PyTorch ops) is complete before k_pack_B reads it, and later default-stream work sees
  // C.  Callers on other streams pass theirs to mxa_dgemm_compressed_device.
  if (!check_hip(hipStreamCreateWithFlags(&h->stream, hipStreamDefault), __func__, __LINE__)) { destroy_handle(h); return 1; }
  // decided above, a property of the object from then on: only the SNP-major copy is staged; plink_transposed is not read
  h->single = single0;
  if (h->single) {
    if (stage_matrix(h->snp_major, plink, plink_pitch, snps, indiv, h->stream)) { destroy_handle(h); return 1; }
    describe_matrix(h->ind_major, indiv, snps);
  } else
  if (one_pointer ? stage_from_snp_major(h, plink, plink_pitch) :
                    (stage_matrix(h->snp_major, plink, plink_pitch, snps, indiv, h->stream) ||
                     stage_matrix(h->ind_major, plink_t, plink_t_pitch, indiv, snps, h->stream))) { destroy_handle(h); return 1; }
  if (!check_hip(hipMalloc(reinterpret_cast<void **>(&h->d_f), sizeof(double) * snps), __func__, __LINE__)) { destroy_handle(h); return 1; }
  h->h_f = (double *)calloc((size_t)snps, sizeof(double));
  if (!h->h_f) { set_error(12, "plink2compressed: out of host memory for %ld allele frequencies", snps); destroy_handle(h); return 1; }
  if (f) {
    hipError_t e = hipMemcpy(h->d_f, f, sizeof(double) * snps, hipMemcpyDefault);
    if (!check_hip(e, __func__, __LINE__)) { destroy_handle(h); return 1; }
    h->has_f = true;
  } else {
    (void)hipMemset(h->d_f, 0, sizeof(double) * snps);
  }
  // (the host copy of the frequencies travels on the object's stream; that alone does NOT pay the runtime's first staged device-to-host copy of a product --
  // measured, "attempt 1" in profiles/r06_harness_phase_clock.txt -- the warm-up products below do)
  if (!check_hip(hipMemcpyAsync(h->h_f, h->d_f, sizeof(double) * snps, hipMemcpyDeviceToHost, h->stream), __func__, __LINE__) ||
      !check_hip(hipStreamSynchronize(h->stream), __func__, __LINE__)) { destroy_handle(h); return 1; }
  if (ensure_workspace(h, h->max_n) || reserve_small_routes(h, h->max_n)) { destroy_handle(h); return 1; }
  if (warm_up(h, src_dev0 < 0)) { destroy_handle(h); return 1; }
  *out = h;
  return 0;
}

// ---- incremental staging (round 5).  plink2compressed wants the whole PLINK matrix behind one pointer; at BASELINE config 4's full extent that is 250 GB
// next to the 250 GB object it becomes.  Here the object is allocated first (ONE packed copy, SNP-major: Handle::single) and filled by blocks of SNP rows
// -- from a generator, a file reader, another device -- so that nothing but the object and one block is ever resident.  The .bed reader uses it for
// single-orientation objects (bed_range_to_handle).
int begin_handle(long snps, long indiv, int max_n, void **out, int device) {
  if (out) *out = nullptr;
  if (!out) { set_error(1, "mxa_plink2compressed_begin: compressed is NULL"); return 1; }
  if (snps <= 0 || indiv <= 0) { set_error(1, "mxa_plink2compressed_begin: snps and indiv must be positive"); return 1; }
  Options &o = options();
  if (!o.set) { o.gpu = true; o.centered = true; o.set = true; }
  if (!o.gpu) { set_error(14, "mxa_plink2compressed_begin: setOptions_compressed was called with use_gpu=0; this library has no CPU engine"); return 1; }
  int dev = device;
  if (dev < 0) dev = select_device();
  else if (!check_hip(hipSetDevice(dev), __func__, __LINE__)) dev = -1;
  if (dev < 0) return 1;
  size_t free_b = 0, total_b = 0;
  MXA_HIP(hipMemGetInfo(&free_b, &total_b));
  // the packed copy + B in fragment order + at least one K split of partial sums (partial_budget cuts the rest into groups) + the staged operands' room
  const size_t need = object_footprint(snps, indiv, max_n, true);
  if (need > free_b) {
    set_error(12, "Not enough device memory available. Required %zu GB, free %zu GB, total on device %zu GB", need >> 30, free_b >> 30, total_b >> 30);
    return 1;
  }
  Handle *h = new Handle();
  h->device = dev; h->snps = snps; h->indiv = indiv; h->max_n = std::max(max_n, 1);
  h->single = true; h->staging = true;
  if (!check_hip(hipStreamCreateWithFlags(&h->stream, hipStreamDefault), __func__, __LINE__)) { destroy_handle(h); return 1; }
  PackedMatrix &M = h->snp_major;
  describe_matrix(M, snps, indiv);
  const size_t bytes = (size_t)M.rows_pad * M.pitch;
  if (!check_hip(hipMalloc(reinterpret_cast<void **>(&M.d), bytes), __func__, __LINE__) ||
      !check_hip(hipMemsetAsync(M.d, 0, bytes, h->stream), __func__, __LINE__)) { destroy_handle(h); return 1; }
  describe_matrix(h->ind_major, indiv, snps);
  if (!check_hip(hipMalloc(reinterpret_cast<void **>(&h->d_f), sizeof(double) * snps), __func__, __LINE__) ||
      !check_hip(hipMemsetAsync(h->d_f, 0, sizeof(double) * snps, h->stream), __func__, __LINE__)) { destroy_handle(h); return 1; }
  h->h_f = (double *)calloc((size_t)snps, sizeof(double));
  if (!h->h_f) { set_error(12, "mxa_plink2compressed_begin: out of host memory for %ld allele frequencies", snps); destroy_handle(h); return 1; }
  h->has_f = true;
  if (ensure_workspace(h, h->max_n)) { destroy_handle(h); return 1; }
  *out = h;
  return 0;
}

int append_rows(Handle *h, const uint8_t *rows, long snp_begin, long nrows, const double *f_rows) {
  if (!h->staging) { set_error(19, "mxa_plink2compressed_rows: the object is sealed (or was not made by mxa_plink2compressed_begin)"); return 1; }
  if (!rows || snp_begin < 0 || nrows <= 0 || snp_begin + nrows > h->snps) {
    set_error(1, "mxa_plink2compressed_rows: need rows != NULL and 0 <= snp_begin, snp_begin + nrows <= snps (got [%ld, %ld) of %ld)", snp_begin, snp_begin + nrows, h->snps);
    return 1;
  }
  // coverage is an interval set, not a count: a block appended twice (a retried block) or overlapping another one is refused, so that an object can only
  // be sealed with every SNP row written exactly once
  auto at = std::lower_bound(h->staged_iv.begin(), h->staged_iv.end(), std::make_pair(snp_begin, snp_begin));
  if ((at != h->staged_iv.end() && at->first < snp_begin + nrows) || (at != h->staged_iv.begin() && std::prev(at)->second > snp_begin)) {
    set_error(1, "mxa_plink2compressed_rows: rows [%ld, %ld) overlap rows that were already appended", snp_begin, snp_begin + nrows);
    return 1;
  }
  MXA_HIP(hipSetDevice(h->device));
  hipStream_t s = h->stream;
  const long bps = (h->indiv + 3) / 4;
  int src_dev = -1;
  const bool on_dev = ptr_location(rows, &src_dev) == 1;
  if (f_rows) MXA_HIP(hipMemcpyAsync(h->d_f + snp_begin, f_rows, sizeof(double) * nrows, hipMemcpyDefault, s));
  if (on_dev) {
    if (src_dev != h->device) {
      if (!enable_peer(h->device, src_dev)) { set_error(15, "mxa_plink2compressed_rows: device %d cannot read the rows in the memory of device %d", h->device, src_dev); return 1; }
      if (sync_foreign_producer(src_dev)) return 1;
    }
    if (launch_recode(rows, (size_t)bps, snp_begin, nrows, h->indiv, 0, h->snp_major, s)) return 1;
    if (!f_rows && launch_allele_freq(rows, nrows, h->indiv, h->d_f + snp_begin, s)) return 1;
    MXA_HIP(hipStreamSynchronize(s));   // the caller may reuse its block buffer
  } else {
    const size_t chunk_bytes = (size_t)256 << 20;
    const long chunk_rows = std::min<long>(nrows, std::max<long>(1, (long)(chunk_bytes / (size_t)bps)));
    uint8_t *bounce = nullptr;
    MXA_HIP(hipMalloc(reinterpret_cast<void **>(&bounce), (size_t)chunk_rows * bps));
    int rc = 0;
    for (long r0 = 0; r0 < nrows && !rc; r0 += chunk_rows) {
      const long nr = std::min(chunk_rows, nrows - r0);
      if (!check_hip(hipMemcpyAsync(bounce, rows + (size_t)r0 * bps, (size_t)nr * bps, hipMemcpyHostToDevice, s), __func__, __LINE__)) { rc = 1; break; }
      rc = launch_recode(bounce, (size_t)bps, snp_begin + r0, nr, h->indiv, 0, h->snp_major, s);
      if (!rc && !f_rows) rc = launch_allele_freq(bounce, nr, h->indiv, h->d_f + snp_begin + r0, s);
      if (!rc && !check_hip(hipStreamSynchronize(s), __func__, __LINE__)) rc = 1;
    }
    (void)hipFree(bounce);
    if (rc) return 1;
  }
  h->staged_rows += nrows;
  h->staged_iv.insert(at, {snp_begin, snp_begin + nrows});
  return 0;
}

int end_handle(Handle *h) {
  if (!h->staging) { set_error(19, "mxa_plink2compressed_end: the object is already sealed"); return 1; }
  long covered = 0, next = 0;          // the intervals are disjoint and sorted: full coverage = they chain from 0 to snps
  for (const auto &iv : h->staged_iv) { if (iv.first != next) break; next = iv.second; covered = next; }
  if (covered != h->snps) {
    set_error(1, "mxa_plink2compressed_end: %ld SNP rows were appended, the object has %ld (first row not yet appended: %ld)", h->staged_rows, h->snps, covered);
    return 1;
  }
  h->staged_iv.clear(); h->staged_iv.shrink_to_fit();
  MXA_HIP(hipSetDevice(h->device));
  MXA_HIP(hipMemcpyAsync(h->h_f, h->d_f, sizeof(double) * h->snps, hipMemcpyDeviceToHost, h->stream));   // (on the object's stream: see create_handle)
  MXA_HIP(hipStreamSynchronize(h->stream));
  h->staging = false;
  if (reserve_small_routes(h, h->max_n)) return 1;
  return warm_up(h, false);
}


// Column-major block copy (height columns of `width` bytes, pitches in bytes) between host / device memory: one hipMemcpy2DAsync (a strided
// 100 MB download takes 1.8 ms this way and 7.7 ms as one 1-D copy per column).  Exception: strided DOWNLOADS issued by the shard worker
// threads of a multi-device object go column by column.  hipMemcpy2DAsync to pageable host memory issued from several threads at the same
// time leaves device memory behind on ROCm 7.2 (0.2-0.6 MiB per shard and object life cycle; not with HIP_LAUNCH_BLOCKING=1, not from a
// single thread, not with 1-D copies: tools/soak_lifecycle.py), and a long-lived session must not creep.
static thread_local bool tl_concurrent = false;
void mark_thread_concurrent() { tl_concurrent = true; }
static hipError_t copy_columns(void *dst, size_t dpitch, const void *src, size_t spitch, size_t width, size_t height, hipStream_t s) {
  if (width == 0 || height == 0) return hipSuccess;
  if (dpitch == width && spitch == width) return hipMemcpyAsync(dst, src, width * height, hipMemcpyDefault, s);
  if (height == 1) return hipMemcpyAsync(dst, src, width, hipMemcpyDefault, s);
  const bool per_column = tl_concurrent && ptr_location(dst, nullptr) == 0;
  if (!per_column) return hipMemcpy2DAsync(dst, dpitch, src, spitch, width, height, hipMemcpyDefault, s);
  for (size_t j = 0; j < height; j++) {
    const hipError_t e = hipMemcpyAsync(static_cast<char *>(dst) + j * dpitch, static_cast<const char *>(src) + j * spitch, width, hipMemcpyDefault, s);
    if (e != hipSuccess) return e;
  }
  return hipSuccess;
}

// ------------------------------------------------------------------------------------------------ multiply
static bool elapsed_ms(hipEvent_t a, hipEvent_t b, float *ms) {
  if (hipEventSynchronize(b) != hipSuccess || hipEventElapsedTime(ms, a, b) != hipSuccess) { (void)hipGetLastError(); return false; }
  return true;
}
static void harvest_slot(Handle *h, int slot) {
  if (!h->prof_pending[slot]) return;
  h->prof_pending[slot] = false;
  float ms = 0.f;
  if (!elapsed_ms(h->ev0[slot], h->ev1[slot], &ms)) return;
  std::lock_guard<std::mutex> lk(g_prof_mutex);
  profile().launches += 1; profile().total_ms += ms;
  h->prof.launches += 1; h->prof.kernel_ms += ms;
  h->prof_last_kernel_ms = ms;
}
static void harvest_copies(Handle *h) {
  float ms = 0.f;
  if (h->in_pending) { h->in_pending = false; if (elapsed_ms(h->ev_in[0], h->ev_in[1], &ms)) { h->prof.in_copies += 1; h->prof.in_ms += ms; h->prof_last_in_ms = ms; } }
  if (h->out_pending) { h->out_pending = false; if (elapsed_ms(h->ev_out[0], h->ev_out[1], &ms)) { h->prof.out_copies += 1; h->prof.out_ms += ms; h->prof_last_out_ms = ms; } }
}
void harvest_profile(Handle *h) {
  if (!h) return;
  harvest_slot(h, 0); harvest_slot(h, 1);
  harvest_copies(h);
}

int sync_foreign_producer(int src_dev) {
  int cur = 0;
  MXA_HIP(hipGetDevice(&cur));
  if (src_dev < 0 || src_dev == cur) return 0;
  MXA_HIP(hipSetDevice(src_dev));
  const hipError_t e = hipStreamSynchronize(nullptr);
  MXA_HIP(hipSetDevice(cur));
  MXA_HIP(e);
  return 0;
}

int enable_peer(int cur, int peer) {
  if (cur == peer) return 1;
  int prev = 0, can = 0;
  if (hipGetDevice(&prev) != hipSuccess) { (void)hipGetLastError(); return 0; }
  if (hipSetDevice(cur) != hipSuccess) { (void)hipGetLastError(); return 0; }
  if (hipDeviceCanAccessPeer(&can, cur, peer) != hipSuccess) { (void)hipGetLastError(); can = 0; }
  if (can) {
    const hipError_t e = hipDeviceEnablePeerAccess(peer, 0);
    if (e != hipSuccess) (void)hipGetLastError();
    if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) can = 0;
  }
  (void)hipSetDevice(prev);
  return can;
}

// Which stored orientation feeds a k_gemm launch (round 4).  Every product can be computed from either copy: in the plain form from the copy whose ROWS
// are the output rows ('T': SNP-major, 'N': individual-major -- the reference's choice, dgemm_compressed_cuda.cu:270), in the transposed-operand form
// (k_gemm<..., TR>) from the copy whose rows are the K index.  The two forms give bit-identical results (same plan, same K order).  The transposed form
// reads the genotype operand with ONE LDS instruction per K-step instead of A / 2 and extracts every fragment with a two-operand v_and_b32 whose scale
// belongs to the output row (undone exactly in the epilogue), for every tile.  Measured, same box, alternating (profiles/r04_gemm_tr_ab.txt): C2
// 42.5 ms per launch against 43.9-44.0 = 75.3 against 72.8 TFLOP/s = 0.957 against 0.925 of the fp64 MFMA peak; config-4 shard 423.6 against 438.0 ms
// (0.961); n = 20: 13.55 against 13.98; n = 16: 10.87 against 11.03; n = 12: 8.30-8.35 against 8.44-8.47; n = 8: 5.70 against 5.77-5.84.
// MXA_GEMM_TR: 0 never (the plain form), otherwise / unset: always.  Read per call.
static bool gemm_use_tr(const GemmPlan &, const Handle *h, bool trans) {
  if (h->single) return !trans;            // one stored copy (SNP-major): 'N' transposed, 'T' plain
  const char *e = getenv("MXA_GEMM_TR");
  return !e || atoi(e) != 0;
}
static const PackedMatrix &gemm_operand(const Handle *h, bool trans, bool tr) { return (trans != tr) ? h->snp_major : h->ind_major; }

constexpr int kSmallNMaxColsHost = 6;   // widest product that takes the guarded exact int8 route whole (mxa_gemm_i8.hip: kSmallNMaxCols)

// Device operands only; asynchronous on s.  With timing, ev0/ev1 bracket the dominant kernel and harvest_profile() reads them later.
static int gemm_device(Handle *h, bool trans, int n, const double *dB, long ldb, double *dC, long ldc, long fill_rows, hipStream_t s, bool timing = true) {
  const PackedMatrix &G = trans ? h->snp_major : h->ind_major;   // reference picks d_plink for 'T' (dgemm_compressed_cuda.cu:270)
  const long m = G.rows, k = G.k;
  const bool centered = options().centered;
  if (centered && !h->has_f) { set_error(6, "dgemm_compressed: centring requested but no allele frequencies were supplied to plink2compressed"); return 1; }
  if (ldb < k || ldc < m) { set_error(7, "dgemm_compressed: leading dimension too small (ldb %ld < %ld or ldc %ld < %ld)", ldb, k, ldc, m); return 1; }
  if (fill_rows < m || fill_rows > ldc) { set_error(7, "internal: fill_rows %ld outside [%ld, %ld]", fill_rows, m, ldc); return 1; }
  if (n > 65535) { set_error(7, "dgemm_compressed: n = %d exceeds the supported 65535 columns per call", n); return 1; }
  if (n > h->max_n) { h->max_n = n; }
  if (ensure_workspace(h, n)) return 1;
  clock_mark("workspace");
  Workspace &w = h->ws;
  double *d_sumB = w.d_colpart + (size_t)n * 128, *d_sumfB = d_sumB + n;
  const int engine = g_engine.load();
  const bool prof = g_profile_on && timing;
  const int slot = h->prof_slot;
  if (prof) {
    if (!h->ev0[slot]) { MXA_HIP(hipEventCreate(&h->ev0[slot])); MXA_HIP(hipEventCreate(&h->ev1[slot])); }
    harvest_slot(h, slot);   // the pair of the product before last is read before it is recorded again
    h->prof_slot = slot ^ 1;
    clock_mark("prof-events");
  }
  hipEvent_t pe0 = prof ? h->ev0[slot] : nullptr, pe1 = prof ? h->ev1[slot] : nullptr;
  // ---- the guarded exact int8 route of narrow products and of peeled columns (HBM-bound).
  // B is split exactly into balanced radix-256 digits and multiplied on the int8 matrix cores with exact integer sums WHEN THAT IS EXACT: every column
  // finite, its exponent span within the digits (e_max - e_min <= 8 S - 55), no underflow in the recombination.  Then no bit of B is dropped and the
  // only roundings are the S - 1 additions of the recombination: |error| <= 3.02 (S - 1) 2^-53 sum_k |z_k b_k|, below the K 2^-53 sum |z b| of an fp64
  // chain for K >= 128.  The verdict is formed ON THE DEVICE (round 5: for every n, not only n <= 2) as a class -- class 0: exact with the digits of the
  // cheapest tile count, class 1: exact with one more tile of digits, class 2: not exact -- and the chains of both classes plus the fp64 kernel behind
  // them are all enqueued, each testing one flag word: NO product waits for the host (rounds 3-4 read three integers back for 3 <= n <= 6 and peeled
  // columns).  Digits per column by n: what fits the tiles of 32 expanded columns -- n = 1: 32; 2: 16; 3: 10 | 21; 4: 16 | 24; 5: 12 | 19; 6: 10 | 16.
  // Single-orientation object, 'N': the plain int8 kernel needs the individual-major copy; k_gemm_i8_tn multiplies from the SNP-major one, one pass per
  // tile of 32 expanded columns; a class whose passes would cost more than the fp64 MFMA tile is not offered (gemm_i8_reserve returns 2).
  const bool no_plain = h->single && !trans;
  const PackedMatrix *G_tn_single = no_plain ? &gemm_operand(h, trans, true) : nullptr;
  const bool small_ok = (engine == 0 || (engine == 4 && n <= 2)) && k >= 128;
  // columns [c0, c0 + nc), nc <= 6.  0: enqueued (int8 chains + fp64 kernel: the columns are done whatever the verdict), 2: not applicable, 1: error
  // Verdict class 2 ("not exactly representable"; inf / NaN; a column near the underflow threshold): who does the product in fp64?
  //   nc = 1: fp64 chains, one thread per output row, inside the first chain's k_slice_B launch (no launch of their own: the CG step counts its launches; with 32 digits =
  //           201 binades the class needs inf / NaN or an entry 60 decades below its column's maximum).
  //   nc >= 2: the plain-operand fp64 path behind the chains, every launch gated by the verdict word -- k_lut + k_finish (nc = 2, plain form) or k_pack_B + k_gemm<MODE 0>
  //           + k_finish: three empty launches (~5 us each) on a >= 1 ms product when the int8 classes apply, against 45-280 ms of chains when they do not
  //           (500k x 50k, n = 4 .. 6: measured late in round 5).  If its partial sums do not fit the budget (objects that fill the device): the chains.
  auto guarded_small = [&](int c0, int nc, const PackedMatrix *G_tn, hipEvent_t e0, hipEvent_t e1, int *splits_out, const int **flag_ptr) -> int {
    I8Chain ch; ch.S0 = kSmallS0[nc]; ch.S1 = kSmallS1[nc];
    const bool fb_lut = nc <= 2 && !no_plain;
    GemmPlan pf = fb_lut ? plan_lut(m, G.k_pad, nc) : plan_gemm(m, G.k_pad, nc);
    bool fast_fb = nc >= 2;
    if (fast_fb) {
      const size_t one = (size_t)pf.n_pad * pf.m_pad, need = one * pf.splits;
      fast_fb = partial_budget(w, need, one) >= need;
    }
    ch.fp64_rows = !fast_fb;
    const int r0 = gemm_i8_reserve(G, nc, ch.S0, G_tn, w, s);
    if (r0) return r0;
    if (ch.S1) {
      const int r1 = gemm_i8_reserve(G, nc, ch.S1, G_tn, w, s);
      if (r1 == 1) return 1;
      if (r1 == 2) ch.S1 = 0;   // the larger class is not offered (transposed form: more passes than the fp64 kernel is worth)
    }
    const double *dBc = dB + (size_t)c0 * ldb;
    double *dCc = dC + (size_t)c0 * ldc;
    const int *d_flag = nullptr;
    for (int cls = 0; cls < (ch.S1 ? 2 : 1); cls++) {
      ch.my_class = cls; ch.first = cls == 0;
      const int rc = gemm_i8_device(G, trans, nc, dBc, ldb, dCc, ldc, fill_rows, centered, d_sumB + c0, d_sumfB + c0, h->d_f, w, s, cls == 0 ? e0 : nullptr, cls == 0 ? e1 : nullptr,
                                    cls == 0 ? splits_out : nullptr, 2, &d_flag, w.d_colpart, 0, G_tn, nullptr, &ch);
      if (rc != 3) return 1;
    }
    if (fast_fb) {   // class 2: the gated fp64 launches (d_flag = the word "class == 2")
      if (ensure_partials(w, pf, s)) return 1;
      if (fb_lut) { if (launch_lut(G, dBc, ldb, nc, w.d_P, pf, s, d_flag)) return 1; }
      else {
        const bool trf = gemm_use_tr(pf, h, trans);
        if (launch_pack_B(dBc, ldb, k, nc, w.d_Bp, G.k_pad, pf.n_pad, pf.c, s, nullptr, 0, -1, d_flag)) return 1;
        if (launch_gemm(gemm_operand(h, trans, trf), w.d_Bp, w.d_P, pf, 0, s, next_ctr(w), 0, -1, d_flag, trf)) return 1;
      }
      if (launch_finish(w.d_P, pf, m, nc, dCc, ldc, fill_rows, trans ? 1 : 0, centered, d_sumB + c0, d_sumfB + c0, h->d_f, s, nullptr, 0, 0, d_flag, nullptr)) return 1;
    }
    // (class 2 at nc = 1: plain fp64 chains, one thread per output row, inside the first chain's k_slice_B launch)
    if (flag_ptr) *flag_ptr = d_flag;
    return 0;
  };
  if (small_ok && n <= kSmallNMaxColsHost) {
    const char *e_tn = getenv("MXA_I8_TN");   // A/B (read per call): n <= 2 from the copy whose rows are the K index
    const bool tn_ab = e_tn && atoi(e_tn) != 0;
    const PackedMatrix *G_tn = no_plain ? G_tn_single : (tn_ab && n <= 2 && !h->single) ? &gemm_operand(h, trans, true) : nullptr;   // (the A/B needs both stored copies)
    int splits8 = 1;
    const int *d_flag = nullptr;
    const int rcx = guarded_small(0, n, G_tn, pe0, pe1, &splits8, &d_flag);
    if (rcx == 1) return 1;
    if (rcx == 0) {   // (the range flag of the denormal-operand mode -- mxa_last_range_fallback -- is cleared by the chain's k_slice_B)
      std::lock_guard<std::mutex> lk(g_prof_mutex);
      Geometry &geo = last_geometry();
      geo.m = m; geo.k = k; geo.n = n; geo.splits = splits8; geo.a = 0; geo.c = 0; geo.path = 4; geo.d_flag = d_flag; geo.flag_dev = h->device;
      h->prof_pending[slot] = prof;
      return 0;
    }
  }
  if (centered && launch_colsums(dB, ldb, k, n, trans ? nullptr : h->d_f, w.d_colpart, d_sumB, d_sumfB, s)) return 1;
  // Engines 1 / 4 (opt-in): the int8 slicing of all n columns with S digits each (0: plan_i8's choice).  Plain form: one call.  Transposed-operand form ('N' of a
  // one-copy object): k_gemm_i8_tn takes at most six tiles of 32 expanded columns per call (three passes of two tiles), so a wide product goes in balanced column
  // chunks -- every pass streams the packed matrix once, and three columns' worth of fp64 MFMA time buys a pass (C2, n = 32, 10 digits: 6 passes of 2.9 ms against
  // 43 ms on the fp64 tile).  stats_base: column maxima that are already there (k_colmax_partial's layout: the maxima of column j at [64 j, 64 j + 64)).
  auto i8_engine_product = [&](int S, double *stats_base, hipEvent_t e0, hipEvent_t e1, int *splits_out) -> int {
    const int rc = gemm_i8_device(G, trans, n, dB, ldb, dC, ldc, fill_rows, centered, d_sumB, d_sumfB, h->d_f, w, s, e0, e1, splits_out, 0, nullptr, nullptr, S, G_tn_single, stats_base);
    if (rc != 2 || !G_tn_single) return rc;
    const int cmax = (6 * 32) / (S > 0 ? S : 7);
    if (cmax < 1) return 2;
    const int chunks = (n + cmax - 1) / cmax, per = (n + chunks - 1) / chunks;
    if (chunks < 2) return 2;                  // declined for another reason than the tile count (K beyond the accumulators' range)
    for (int c0 = 0; c0 < n; c0 += per) {
      const int nc = std::min(per, n - c0);
      const int rcc = gemm_i8_device(G, trans, nc, dB + (size_t)c0 * ldb, ldb, dC + (size_t)c0 * ldc, ldc, fill_rows, centered, d_sumB + c0, d_sumfB + c0, h->d_f, w, s,
                                     c0 == 0 ? e0 : nullptr, c0 + nc >= n ? e1 : nullptr, splits_out, 0, nullptr, nullptr, S, G_tn_single, stats_base ? stats_base + (size_t)c0 * 64 : nullptr);
      if (rcc == 2 && c0 == 0) return 2;       // nothing is enqueued yet: the fp64 path takes the product
      if (rcc) { if (rcc == 2) set_error(4, "internal: a later column chunk of the int8 engine declined"); return 1; }
    }
    return 0;
  };
  // Engine 4 (i8-exact, opt-in): the same exact slicing for EVERY n with the digit count chosen PER CALL from the measured exponent span of B's columns --
  // S = max(7, ceil((span + 55) / 8)) <= 24 -- by the host: three integers are read back (ONE host synchronisation per call, documented with the engine).
  if (engine == 4 && k >= 128 && n > 2) {
    int splits8 = 1, hs[3] = {0, 0, 1};
    if (launch_colspan(dB, ldb, k, n, w.d_colpart, w.d_denflag + 4, s)) return 1;
    MXA_HIP(hipMemcpyAsync(hs, w.d_denflag + 4, sizeof(hs), hipMemcpyDeviceToHost, s));
    MXA_HIP(hipStreamSynchronize(s));
    const int S = std::max(7, (hs[0] + 55 + 7) / 8);
    const int rcx = (hs[2] || S > kI8ExactMaxDigits || hs[1] < 8 * S - 1023) ? 2 : i8_engine_product(S, w.d_colpart, pe0, pe1, &splits8);
    if (rcx == 1) return 1;
    if (rcx == 0) {
      MXA_HIP(hipMemsetAsync(w.d_denflag, 0, sizeof(int), s));
      std::lock_guard<std::mutex> lk(g_prof_mutex);
      Geometry &geo = last_geometry();
      geo.m = m; geo.k = k; geo.n = n; geo.splits = splits8; geo.a = S; geo.c = 0; geo.path = 2; geo.d_flag = nullptr; geo.flag_dev = h->device;
      h->prof_pending[slot] = prof;
      return 0;
    }
  }
  if (engine == 1) {   // opt-in: the int8 slicing without the exactness check (7 digits; 32 / 16 for n = 1 / 2)
    int splits8 = 1;
    const int rc8 = i8_engine_product(0, nullptr, pe0, pe1, &splits8);
    if (rc8 == 0) {
      std::lock_guard<std::mutex> lk(g_prof_mutex);
      Geometry &geo = last_geometry();
      geo.m = m; geo.k = k; geo.n = n; geo.splits = splits8; geo.a = 0; geo.c = 0; geo.path = 2; geo.d_flag = nullptr; geo.flag_dev = h->device;
      h->prof_pending[slot] = prof;
      return 0;
    }
    if (rc8 != 2) return 1;   // 2: the int8 route declined (K beyond the accumulators' range in the transposed-operand form): fp64 path below
  }
  // Column peel (engine 0, n = 4q + r > 6, r = 1, 2, 3): the MFMA tile works on groups of 4 columns, so 10 columns would cost 12 (the reference harness's
  // default n = 10).  The r odd columns go through the guarded route above -- one HBM-bound pass over the packed matrix -- and the multiple of 4 runs on the
  // fp64 MFMA without padding.
  const int n_odd = n & 3;
  if (small_ok && n > kSmallNMaxColsHost && n_odd != 0) {
    const int rcp = guarded_small(n - n_odd, n_odd, G_tn_single, nullptr, nullptr, nullptr, nullptr);
    if (rcp == 1) return 1;
    if (rcp == 0) n -= n_odd;            // the rest of this function multiplies the first 4q columns
    clock_mark("peel-enqueued");
  }
  const bool use_lut = n <= 2 && !no_plain;   // fp64 pair tables: engine f64-strict, and K < 128
  GemmPlan p = use_lut ? plan_lut(m, G.k_pad, n) : plan_gemm(m, G.k_pad, n);
  // K splits per launch group: all of them unless their partial sums exceed the budget (partial_budget)
  int splits_per_group = p.splits;
  if (!use_lut) {
    const size_t one = (size_t)p.n_pad * p.m_pad, need = one * p.splits;
    const size_t budget = partial_budget(w, need, one);
    if (budget < need) {
      splits_per_group = (int)std::max<size_t>(1, budget / one);
      GemmPlan pg = p; pg.splits = splits_per_group;
      if (ensure_partials(w, pg, s)) return 1;
    }
  }
  if (splits_per_group == p.splits && ensure_partials(w, p, s)) return 1;   // the plan of the columns left after a peel may need more than the plan ensure_workspace sized for
  clock_mark("partials");
  {
    std::lock_guard<std::mutex> lk(g_prof_mutex);
    Geometry &geo = last_geometry();
    geo.m = m; geo.k = k; geo.n = n; geo.splits = p.splits; geo.a = p.a; geo.c = p.c; geo.path = use_lut ? 1 : 0;
  }
  // MODE 2 / 3 (default): genotype operand as the denormal z * 2^-1074 (one VALU per fragment instead of two); B scaled per column
  const bool tr = !use_lut && gemm_use_tr(p, h, trans);   // transposed-operand form: from the OTHER stored orientation (gemm_use_tr)
  const PackedMatrix &GL = gemm_operand(h, trans, tr);
  int mode = gemm_default_mode(p.c);
  const int *d_E = nullptr;
  if (!use_lut && (mode == 2 || mode == 3)) {
    if (w.cap_exp < (size_t)n) {
      MXA_HIP(hipStreamSynchronize(s));
      if (w.d_exp) { MXA_HIP(hipFree(w.d_exp)); w.d_exp = nullptr; w.cap_exp = 0; }
      MXA_HIP(hipMalloc(reinterpret_cast<void **>(&w.d_exp), sizeof(int) * (size_t)n));
      w.cap_exp = n;
    }
    // per-column exponents + the range guard of the mode: a column whose non-zero entries span more than kDenMaxSpan binades (or that
    // holds inf / NaN) raises d_denflag, and the plain-operand chain below redoes the product (the verdict stays on the device)
    if (launch_colexp(dB, ldb, k, n, w.d_colpart, w.d_exp, 0, s, w.d_denflag, kDenMaxSpan, -100000)) return 1;
    d_E = w.d_exp;
  }
  if (!use_lut && launch_pack_B(dB, ldb, k, n, w.d_Bp, G.k_pad, p.n_pad, p.c, s, d_E, 0, -1, nullptr, mode == 3 && !tr)) return 1;
  if (prof) MXA_HIP(hipEventRecord(pe0, s));
  if (splits_per_group < p.splits) {
    // GROUPED K splits (round 5; BASELINE config 4 at its full 5M x 200k x 128 on one device): the launch plan is unchanged -- same pieces, same K
    // order -- but only `splits_per_group` splits are in flight at a time; each group's partial sums are added to the running sum kept in C (raw: no
    // scale-back, no centring) and the last group applies the epilogue.  One sequential ascending chain of additions, as in the one-pass k_finish:
    // bit-identical to it (tests/test_grouped_splits_gpu.py).  Pass 1 = the plain-operand fallback of the denormal-operand mode, gated by the
    // range flag like the one-pass path: it redoes every group and overwrites C.
    const size_t stride = (size_t)p.n_pad * p.m_pad;
    for (int pass = 0; pass < (d_E ? 2 : 1); pass++) {
      const int *gate = pass ? w.d_denflag : nullptr;
      if (pass && launch_pack_B(dB, ldb, k, n, w.d_Bp, G.k_pad, p.n_pad, p.c, s, nullptr, 0, -1, gate)) return 1;
      for (int sb = 0; sb < p.splits; sb += splits_per_group) {
        const int se = std::min(p.splits, sb + splits_per_group);
        // split sb lands at the start of the buffer (p_split0 = sb)
        if ((size_t)(se - sb) * stride > w.cap_P) { set_error(4, "internal: a group of %d K splits exceeds the partial-sum workspace", se - sb); return 1; }
        if (launch_gemm(GL, w.d_Bp, w.d_P, p, pass ? 0 : mode, s, next_ctr(w), sb, se, gate, tr, sb)) return 1;
        const int group = (sb > 0 ? 1 : 0) | (se < p.splits ? 2 : 0);
        if (launch_finish(w.d_P, p, m, n, dC, ldc, fill_rows, trans ? 1 : 0, centered, d_sumB, d_sumfB, h->d_f, s, pass ? nullptr : d_E, 0, 0, gate, nullptr, se - sb, group)) return 1;
      }
      if (!pass && prof) { MXA_HIP(hipEventRecord(pe1, s)); h->prof_pending[slot] = true; }
    }
    return 0;
  }
  int rc = use_lut ? launch_lut(G, dB, ldb, n, w.d_P, p, s) : launch_gemm(GL, w.d_Bp, w.d_P, p, mode, s, next_ctr(w), 0, -1, nullptr, tr);
  if (prof && !rc) { MXA_HIP(hipEventRecord(pe1, s)); h->prof_pending[slot] = true; }
  if (!rc && d_E) {   // fallback of the denormal-operand mode, run only when the guard raised the flag: unscaled B, two-instruction conversion
    rc = launch_pack_B(dB, ldb, k, n, w.d_Bp, G.k_pad, p.n_pad, p.c, s, nullptr, 0, -1, w.d_denflag);
    if (!rc) rc = launch_gemm(GL, w.d_Bp, w.d_P, p, 0, s, next_ctr(w), 0, -1, w.d_denflag, tr);
  }
  if (!rc) rc = launch_finish(w.d_P, p, m, n, dC, ldc, fill_rows, trans ? 1 : 0, centered, d_sumB, d_sumfB, h->d_f, s, d_E, 0, 0, nullptr, d_E ? w.d_denflag : nullptr);
  return rc;
}


// ------------------------------------------------------------------------------------------------ host-operand pipeline
// The plain reference ABI hands over HOST B and C (utils/benchmark/benchmark.f90:192-209 times exactly that).  At C2 a call moves 269 MB over
// PCIe (5.4 ms at the ~50 GB/s a pageable copy reaches) around a 44 ms product.  Both transfers are hidden behind the product:
//   big B (K-chunk mode, 'N'): the K splits of the launch plan are cut into <= 8 groups; group c's rows of B are uploaded (the host
//     thread blocks in the staged pageable copy while the GPU multiplies group c-1), scaled and packed on their own (per-group column
//     exponents), and multiplied by a launch over just those splits.  Groups alternate between two streams so that the ramp of one launch
//     fills the drain of the previous one.  One k_finish adds all partials in the usual ascending order, each scaled back by its group's
//     exponent first -- power-of-two scaling commutes with rounding, so the result is bit-identical to the one-launch path.
//   big C (M-chunk mode, 'T'): B is uploaded and packed once; <= 8 row ranges of the packed matrix are multiplied and finished by launches of
//     their own (same split count as the one-launch plan, so identical sums), alternating between the two streams, and every finished row
//     range travels to the host while the next one is computed.
constexpr int kPipeChunks = 8;            // at most; at least ~8 MB of transfer per chunk
static const size_t kPipeMinBytes = (size_t)32 << 20;

static int pipe_setup(Handle *h) {
  for (hipStream_t &ps : h->pipe) if (!ps) MXA_HIP(hipStreamCreateWithFlags(&ps, hipStreamNonBlocking));
  for (hipEvent_t &e : h->pev) if (!e) MXA_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  return 0;
}

// returns 0 done, 1 error, 2 not applicable (caller takes the plain path)
static int gemm_host_pipelined(Handle *h, bool trans, int n, const double *B, long ldb, bool b_host, bool b_local, double *C, long ldc, bool c_host, bool c_local,
                               long fill_rows) {
  const int engine = g_engine.load();
  if ((engine != 0 && engine != 3) || n < 3 || getenv("MXA_DIAG")) return 2;
  if (engine == 0 && n <= kSmallNMaxColsHost) return 2;   // engine 0 sends n <= 6 through the guarded int8 route of gemm_device (plain upload; B is at most 6 columns)
  const PackedMatrix &G = trans ? h->snp_major : h->ind_major;
  const long m = G.rows, k = G.k;
  // b_host / c_host: the operand is not memory of this device -- host memory (PCIe) or memory of another GPU (peer copies over xGMI):
  // either way its transfer is hidden behind the product
  const size_t b_bytes = b_host ? sizeof(double) * (size_t)k * n : 0, c_bytes = c_host ? sizeof(double) * (size_t)m * n : 0;
  if (std::max(b_bytes, c_bytes) < kPipeMinBytes) return 2;
  const bool kmode = b_bytes >= c_bytes;
  const bool centered = options().centered;
  if (centered && !h->has_f) { set_error(6, "dgemm_compressed: centring requested but no allele frequencies were supplied to plink2compressed"); return 1; }
  if (n > h->max_n) h->max_n = n;
  if (ensure_workspace(h, n) || pipe_setup(h)) return 1;
  harvest_profile(h);
  Workspace &w = h->ws;
  hipStream_t s = h->stream;
  // scratch: per-group column-maximum partials (64 n doubles each) + column sums; per-group exponents
  if (grow(&w.d_colpart, &w.cap_colpart, (size_t)n * (128 * (kPipeChunks + 1) + 2) + 16)) return 1;
  if (w.cap_exp < (size_t)n * kPipeChunks) {
    MXA_HIP(hipStreamSynchronize(s));
    if (w.d_exp) { MXA_HIP(hipFree(w.d_exp)); w.d_exp = nullptr; w.cap_exp = 0; }
    MXA_HIP(hipMalloc(reinterpret_cast<void **>(&w.d_exp), sizeof(int) * (size_t)n * kPipeChunks));
    w.cap_exp = (size_t)n * kPipeChunks;
  }
  double *d_sumB = w.d_colpart + (size_t)n * 128 * (kPipeChunks + 1), *d_sumfB = d_sumB + n;
  double *d_sumscratch = w.d_colpart + (size_t)n * 128 * kPipeChunks;   // 128 n doubles for launch_colsums
  // operands on this device
  const double *dB = B; long dldb = ldb;
  if (!b_local) { if (grow(&w.d_Bstage, &w.cap_Bstage, (size_t)k * n)) return 1; dB = w.d_Bstage; dldb = k; }
  double *dC = C; long dldc = ldc;
  if (!c_local) { if (grow(&w.d_Cstage, &w.cap_Cstage, (size_t)fill_rows * n)) return 1; dC = w.d_Cstage; dldc = fill_rows; }
  // column peel as in gemm_device: the 1-3 odd columns go first (their part of B is uploaded ahead of the pipeline) through the same guarded exact int8
  // route -- gemm_device itself on those columns, so that the host path stays bitwise the device path --; the K-range / row-range pipeline then multiplies
  // the multiple of 4
  const int n_all = n, n_odd = n & 3;
  bool b_uploaded = false;
  if (engine == 0 && n > kSmallNMaxColsHost && n_odd != 0 && k >= 128) {
    const int n4 = n - n_odd;
    if (!b_local) {
      if (kmode) MXA_HIP(copy_columns(w.d_Bstage + (size_t)n4 * k, sizeof(double) * k, B + (size_t)n4 * ldb, sizeof(double) * ldb, sizeof(double) * k, n_odd, s));
      else {   // row-range mode uploads the whole (small) B anyway
        if (ldb == k) MXA_HIP(hipMemcpyAsync(w.d_Bstage, B, sizeof(double) * (size_t)k * n, hipMemcpyDefault, s));
        else MXA_HIP(copy_columns(w.d_Bstage, sizeof(double) * k, B, sizeof(double) * ldb, sizeof(double) * k, n, s));
        b_uploaded = true;
      }
    }
    if (gemm_device(h, trans, n_odd, dB + (size_t)n4 * dldb, dldb, dC + (size_t)n4 * dldc, dldc, fill_rows, s, false)) return 1;
    n = n4;
  }
  const GemmPlan p = plan_gemm(m, G.k_pad, n);
  bool tr = gemm_use_tr(p, h, trans);
  if (tr && !kmode && p.a != 8) {   // row ranges of a transposed launch are column ranges of the packed matrix: whole slabs only for the 128-row blocks of A = 8
    if (h->single) return 2;        // (a single-orientation object has no plain form for 'N': the caller's unpipelined path does this product)
    tr = false;
  }
  const PackedMatrix &GL = gemm_operand(h, trans, tr);
  int mode = gemm_default_mode(p.c);
  if (mode != 2 && mode != 3) return 2;
  { const size_t one = (size_t)p.n_pad * p.m_pad; if (partial_budget(w, one * p.splits, one) < one * p.splits) return 2; }   // partial sums beyond the budget: the grouped path of gemm_device
  if (ensure_partials(w, p, s)) return 1;
  { std::lock_guard<std::mutex> lk(g_prof_mutex); Geometry &geo = last_geometry(); geo.m = m; geo.k = k; geo.n = n; geo.splits = p.splits; geo.a = p.a; geo.c = p.c; geo.path = 0; }
  MXA_HIP(hipMemsetAsync(w.d_denflag, 0, sizeof(int), s));   // range guard of the denormal-operand mode, raised by any K group's launch_colexp
  MXA_HIP(hipEventRecord(h->pev[0], s));
  for (hipStream_t ps : h->pipe) MXA_HIP(hipStreamWaitEvent(ps, h->pev[0], 0));   // earlier calls are done with Bp / P
  const int want_chunks = (int)std::max<size_t>(2, std::min<size_t>(kPipeChunks, std::max(b_bytes, c_bytes) >> 23));
  if (kmode) {
    const int spc = (p.splits + want_chunks - 1) / want_chunks, nch = (p.splits + spc - 1) / spc;
    for (int c = 0; c < nch; c++) {
      const int sb = c * spc, se = std::min(p.splits, sb + spc);
      const long slab0 = plan_split_begin(p, sb), slab1 = std::min<long>(p.slabs_total, plan_split_begin(p, se));
      const long k0 = std::min(k, slab0 * kSlabK), k1 = std::min(k, slab1 * kSlabK);
      if (!b_local && k1 > k0)
        MXA_HIP(copy_columns(w.d_Bstage + k0, sizeof(double) * k, B + k0, sizeof(double) * ldb, sizeof(double) * (k1 - k0), n, s));
      MXA_HIP(hipEventRecord(h->pev[2 + c], s));
      hipStream_t cs = h->pipe[c & 1];
      MXA_HIP(hipStreamWaitEvent(cs, h->pev[2 + c], 0));
      int *d_Ec = w.d_exp + (size_t)c * n;
      if (launch_colexp(dB + k0, dldb, k1 - k0, n, w.d_colpart + (size_t)c * 128 * n, d_Ec, 0, cs, w.d_denflag, kDenMaxSpan, -100000, false)) return 1;
      if (launch_pack_B(dB, dldb, k, n, w.d_Bp, G.k_pad, p.n_pad, p.c, cs, d_Ec, slab0 * kSlabSteps, (slab1 - slab0) * kSlabSteps, nullptr, mode == 3 && !tr)) return 1;
      if (launch_gemm(GL, w.d_Bp, w.d_P, p, mode, cs, next_ctr(w), sb, se, nullptr, tr)) return 1;
      MXA_HIP(hipEventRecord(h->pev[10 + c], cs));
    }
    for (int c = 0; c < nch; c++) MXA_HIP(hipStreamWaitEvent(s, h->pev[10 + c], 0));
    if (centered && launch_colsums(dB, dldb, k, n, trans ? nullptr : h->d_f, d_sumscratch, d_sumB, d_sumfB, s)) return 1;
    if (launch_finish(w.d_P, p, m, n, dC, dldc, fill_rows, trans ? 1 : 0, centered, d_sumB, d_sumfB, h->d_f, s, w.d_exp, spc, n)) return 1;
    if (!c_local) {
      if (ldc == fill_rows) MXA_HIP(hipMemcpyAsync(C, dC, sizeof(double) * (size_t)fill_rows * n_all, hipMemcpyDefault, s));
      else MXA_HIP(copy_columns(C, sizeof(double) * ldc, dC, sizeof(double) * fill_rows, sizeof(double) * fill_rows, n_all, s));
    }
  } else {
    if (!b_local && !b_uploaded) {
      if (ldb == k) MXA_HIP(hipMemcpyAsync(w.d_Bstage, B, sizeof(double) * (size_t)k * n, hipMemcpyDefault, s));
      else MXA_HIP(copy_columns(w.d_Bstage, sizeof(double) * k, B, sizeof(double) * ldb, sizeof(double) * k, n, s));
    }
    if (launch_colexp(dB, dldb, k, n, w.d_colpart, w.d_exp, 0, s, w.d_denflag, kDenMaxSpan, -100000, false)) return 1;
    if (launch_pack_B(dB, dldb, k, n, w.d_Bp, G.k_pad, p.n_pad, p.c, s, w.d_exp, 0, -1, nullptr, mode == 3 && !tr)) return 1;
    if (centered && launch_colsums(dB, dldb, k, n, trans ? nullptr : h->d_f, d_sumscratch, d_sumB, d_sumfB, s)) return 1;
    MXA_HIP(hipEventRecord(h->pev[1], s));
    const long rows_chunk = ((m + want_chunks - 1) / want_chunks + kRowAlign - 1) / kRowAlign * kRowAlign;
    {   // the row ranges pad to whole row blocks each: a few blocks more than the one-launch plan
      size_t total = 0;
      for (long r0 = 0; r0 < m; r0 += rows_chunk) { const GemmPlan pc = plan_gemm(std::min(m, r0 + rows_chunk) - r0, G.k_pad, n, &p); total += (size_t)pc.splits * pc.n_pad * pc.m_pad; }
      if (total > w.cap_P) { MXA_HIP(hipStreamSynchronize(s)); if (grow(&w.d_P, &w.cap_P, total)) return 1; }
    }
    size_t p_off = 0;
    int nch = 0;
    for (long r0 = 0; r0 < m; r0 += rows_chunk, nch++) {
      const int c = nch;
      const long r1 = std::min(m, r0 + rows_chunk), rows_c = r1 - r0;
      const bool last = r1 == m;
      PackedMatrix V = GL;   // output rows [r0, r1): whole 256-row tiles, so the view starts at a tile boundary of the tiled layout
      if (!tr) {
        V.d = G.d + (size_t)(r0 / kTileRows) * G.nslabs * kTileBytes;
        V.rows = rows_c; V.rows_pad = last ? G.rows_pad - r0 : rows_chunk;
      } else V.d = GL.d + (size_t)(r0 / kSlabK) * kTileBytes;   // transposed: the output rows are the packed COLUMNS -- the view starts at slab r0 / 128 of every tile row (same pitch)
      const GemmPlan pc = plan_gemm(rows_c, G.k_pad, n, &p);   // the one-launch plan's K pieces: identical sums
      const size_t p_need = (size_t)pc.splits * pc.n_pad * pc.m_pad;
      if (p_off + p_need > w.cap_P) { set_error(4, "internal: partial-result workspace too small for the row-range pipeline"); return 1; }
      hipStream_t cs = h->pipe[c & 1];
      MXA_HIP(hipStreamWaitEvent(cs, h->pev[1], 0));
      if (launch_gemm(V, w.d_Bp, w.d_P + p_off, pc, mode, cs, next_ctr(w), 0, -1, nullptr, tr)) return 1;
      const long fill_c = last ? fill_rows - r0 : rows_c;
      if (launch_finish(w.d_P + p_off, pc, rows_c, n, dC + r0, dldc, fill_c, trans ? 1 : 0, centered, d_sumB, d_sumfB, (h->d_f && trans) ? h->d_f + r0 : h->d_f, cs, w.d_exp)) return 1;
      MXA_HIP(hipEventRecord(h->pev[10 + c], cs));
      p_off += p_need;
    }
    for (int c = 0; c < nch; c++) {
      const long r0 = (long)c * rows_chunk, r1 = std::min(m, r0 + rows_chunk);
      const long cnt = (r1 == m) ? fill_rows - r0 : r1 - r0;
      MXA_HIP(hipStreamWaitEvent(s, h->pev[10 + c], 0));
      if (!c_local) MXA_HIP(copy_columns(C + r0, sizeof(double) * ldc, dC + r0, sizeof(double) * fill_rows, sizeof(double) * cnt, n_all, s));
    }
  }
  MXA_HIP(hipStreamSynchronize(s));
  // a column outside the range of the denormal-operand mode (kDenMaxSpan; absurd inputs): the plain path redoes the product with its
  // on-device fallback
  int den = 0;
  MXA_HIP(hipMemcpy(&den, w.d_denflag, sizeof(int), hipMemcpyDeviceToHost));
  return den ? 2 : 0;
}

static int refuse_while_staging(const Handle *h, const char *who) {
  if (!h->staging) return 0;
  set_error(19, "%s: the object is still being staged (%ld of %ld SNP rows appended); call mxa_plink2compressed_end first", who, h->staged_rows, h->snps);
  return 1;
}

int gemm_any(Handle *h, bool trans, int n, const double *B, long ldb, double *C, long ldc, long fill_rows, bool sync, bool timing) {
  if (refuse_while_staging(h, "dgemm_compressed")) return 1;
  MXA_HIP(hipSetDevice(h->device));
  const PackedMatrix &G = trans ? h->snp_major : h->ind_major;
  const long m = G.rows, k = G.k;
  if (n <= 0) return 0;
  if (!B || !C) { set_error(1, "dgemm_compressed: B and C must not be NULL"); return 1; }
  if (ldb < k || ldc < m) { set_error(7, "dgemm_compressed: leading dimension too small (ldb %ld < %ld or ldc %ld < %ld)", ldb, k, ldc, m); return 1; }
  hipStream_t s = h->stream;
  // phase clock of the call (PRINT_LEVEL > 0 / print_details): one line per call, see CallClock
  CallClock clk;
  clk.start(sync && env_print_level() > 0);   // (the environment only: the reference harness passes print_details = 1, and its timed loop should not print)
  struct ClockScope {
    CallClock *prev; CallClock &c; bool trans; int n;
    ClockScope(CallClock &c_, bool t, int n_) : prev(tl_call_clock), c(c_), trans(t), n(n_) { tl_call_clock = &c; }
    ~ClockScope() { char head[96]; snprintf(head, sizeof(head), "dgemm_compressed '%c' n=%d%s", trans ? 'T' : 'N', n, tl_in_warmup ? " (warm-up inside plink2compressed)" : ""); c.report(head); tl_call_clock = prev; }
  } clock_scope(clk, trans, n);
  int b_devno = -1, c_devno = -1;
  const bool b_local = ptr_location(B, &b_devno) == 1 && b_devno == h->device;
  const bool c_local = ptr_location(C, &c_devno) == 1 && c_devno == h->device;
  clk.mark("locate");
  if (b_devno >= 0 && !b_local && sync_foreign_producer(b_devno)) return 1;   // B may still be being produced on the other device's default stream
  if (sync && (!b_local || !c_local)) {   // an operand in host memory or on another GPU, synchronous call: transfers hidden behind the product when they are large
    const int rcp = gemm_host_pipelined(h, trans, n, B, ldb, !b_local, b_local, C, ldc, !c_local, c_local, fill_rows);
    if (rcp != 2) { clk.mark("pipelined"); return rcp; }
  }
  const double *dB = B; long dldb = ldb;
  double *dC = C; long dldc = ldc;
  Workspace &w = h->ws;
  if (!b_local || !c_local) {
    harvest_copies(h);
    for (hipEvent_t *e : {&h->ev_in[0], &h->ev_in[1], &h->ev_out[0], &h->ev_out[1]}) if (!*e) MXA_HIP(hipEventCreate(e));
    clk.mark("copy-events");
  }
  if (!b_local) {   // host memory, or memory of another device (peer copy over xGMI): dense k x n copy into this device's staging buffer
    if (grow(&w.d_Bstage, &w.cap_Bstage, (size_t)k * n)) return 1;
    clk.mark("grow-Bstage");
    MXA_HIP(hipEventRecord(h->ev_in[0], s));
    if (ldb == k) MXA_HIP(hipMemcpyAsync(w.d_Bstage, B, sizeof(double) * (size_t)k * n, hipMemcpyDefault, s));
    else MXA_HIP(copy_columns(w.d_Bstage, sizeof(double) * k, B, sizeof(double) * ldb, sizeof(double) * k, n, s));
    MXA_HIP(hipEventRecord(h->ev_in[1], s));
    h->in_pending = true;
    dB = w.d_Bstage; dldb = k;
    clk.mark("upload-B");
  }
  if (!c_local) {
    if (grow(&w.d_Cstage, &w.cap_Cstage, (size_t)fill_rows * n)) return 1;
    dC = w.d_Cstage; dldc = fill_rows;
    clk.mark("grow-Cstage");
  }
  if (gemm_device(h, trans, n, dB, dldb, dC, dldc, fill_rows, s, timing)) return 1;
  clk.mark("enqueued");
  if (!c_local) {
    MXA_HIP(hipEventRecord(h->ev_out[0], s));
    if (ldc == fill_rows) MXA_HIP(hipMemcpyAsync(C, dC, sizeof(double) * (size_t)fill_rows * n, hipMemcpyDefault, s));
    else MXA_HIP(copy_columns(C, sizeof(double) * ldc, dC, sizeof(double) * fill_rows, sizeof(double) * fill_rows, n, s));
    MXA_HIP(hipEventRecord(h->ev_out[1], s));
    h->out_pending = true;
    clk.mark("download-C");
  }
  if (sync) {
    MXA_HIP(hipStreamSynchronize(s));
    clk.mark("stream-wait");
    harvest_profile(h);
    if (clk.on) {   // device-side durations of the same call, from the events that are there anyway
      char tmp[96];
      snprintf(tmp, sizeof(tmp), " | device: kernel %.3f in %.3f out %.3f", h->prof.launches ? h->prof_last_kernel_ms : 0.0, h->prof_last_in_ms, h->prof_last_out_ms);
      if (clk.len < (int)sizeof(clk.line) - 100) clk.len += snprintf(clk.line + clk.len, sizeof(clk.line) - clk.len, "%s", tmp);
    }
  }
  return 0;
}

// out (indiv x n) = Zc * (Zc^T * V): the 'T' then the 'N' product with the snps x n intermediate kept in HBM
int gram_any(Handle *h, int n, const double *V, long ldv, double *out, long ldo, bool sync) {
  if (refuse_while_staging(h, "mxa_gram_matvec")) return 1;
  MXA_HIP(hipSetDevice(h->device));
  const long snps = h->snps, indiv = h->indiv;
  if (n <= 0) return 0;
  if (!V || !out) { set_error(1, "mxa_gram_matvec: V and out must not be NULL"); return 1; }
  if (ldv < indiv || ldo < indiv) { set_error(7, "mxa_gram_matvec: leading dimension too small (ldv %ld, ldo %ld < %ld)", ldv, ldo, indiv); return 1; }
  hipStream_t s = h->stream;
  Workspace &w = h->ws;
  int v_devno = -1, o_devno = -1;
  const bool v_local = ptr_location(V, &v_devno) == 1 && v_devno == h->device;
  const bool o_local = ptr_location(out, &o_devno) == 1 && o_devno == h->device;
  const double *dV = V; long dldv = ldv;
  double *dO = out; long dldo = ldo;
  if (!v_local) {
    if (grow(&w.d_Bstage, &w.cap_Bstage, (size_t)std::max(snps, indiv) * n)) return 1;
    if (v_devno >= 0 && sync_foreign_producer(v_devno)) return 1;
    if (ldv == indiv) MXA_HIP(hipMemcpyAsync(w.d_Bstage, V, sizeof(double) * (size_t)indiv * n, hipMemcpyDefault, s));
    else MXA_HIP(copy_columns(w.d_Bstage, sizeof(double) * indiv, V, sizeof(double) * ldv, sizeof(double) * indiv, n, s));
    dV = w.d_Bstage; dldv = indiv;
  }
  if (!o_local) {
    if (grow(&w.d_Cstage, &w.cap_Cstage, (size_t)ldo * n)) return 1;
    dO = w.d_Cstage; dldo = ldo;
  }
  if (grow(&w.d_tmp, &w.cap_tmp, (size_t)snps * n)) return 1;
  if (gemm_device(h, true, n, dV, dldv, w.d_tmp, snps, snps, s, false)) return 1;
  if (gemm_device(h, false, n, w.d_tmp, snps, dO, dldo, dldo, s, false)) return 1;
  if (!o_local) MXA_HIP(hipMemcpyAsync(out, dO, sizeof(double) * (size_t)ldo * n, hipMemcpyDefault, s));
  if (sync) MXA_HIP(hipStreamSynchronize(s));
  return 0;
}

}  // namespace mxa

using namespace mxa;

// ================================================================================================ C ABI
extern "C" {

void setOptions_compressed(int use_gpu, int cores, int floatLoop, int meanSubstract, int ignore_missings, int do_not_center,
                           int do_normalize, int use_miraculix_freq, int variant, int print_details) {
  (void)cores; (void)floatLoop; (void)meanSubstract; (void)variant;
  if (print_details > 0 || env_print_level() > 0) printf("get started\n");  // the reference prints this unconditionally (5codesAPI.c:56)
  clear_error();
  if (!use_gpu) {
    // GPU-only engine.  The host process (a Julia / R session) must survive: report, remember, and let every later
    // plink2compressed leave its handle NULL (the Julia binding throws on a NULL handle, miraculix.jl:29-35).
    Options &o = options();
    o.gpu = false; o.set = true;
    set_error(14, "setOptions_compressed(use_gpu=0): this library is the MI355X engine only; the CPU 5codes engine is not part of "
                  "it. Objects cannot be created until setOptions_compressed is called with use_gpu=1; load the reference library "
                  "for CPU runs.");
    return;
  }
  // same fatal combination as the reference (5codesChar.cc:192-193, ERR0 -> fprintf(stderr)+exit)
  if (use_miraculix_freq || !ignore_missings || do_normalize) {
    fprintf(stderr, "in case of 'gpu' the Fortran frequency must always be used; missings/centering/normalizing cannot treated.\n");
    exit(EXIT_FAILURE);
  }
  Options &o = options();
  o.gpu = true;
  o.centered = !do_not_center;
  o.print_level = print_details;
  o.set = true;
}

void plink2compressed(char *plink, char *plink_transposed, int snps, int indiv, double *f, int max_n, void **compressed) {
  clear_error();
  const size_t ps = ((size_t)indiv + 3) / 4, pi = ((size_t)snps + 3) / 4;
  const int shards = multi_requested();
  if (shards > 1 || getenv("MXA_FORCE_MULTI")) {   // MIRACULIX_NUM_GPUS > 1: SNP blocks over several devices behind the same handle
    (void)multi_create(reinterpret_cast<const uint8_t *>(plink), reinterpret_cast<const uint8_t *>(plink_transposed), snps, indiv, f, max_n, shards, compressed);
    return;
  }
  (void)create_handle(reinterpret_cast<const uint8_t *>(plink), ps, reinterpret_cast<const uint8_t *>(plink_transposed), pi, snps, indiv, f,
                      max_n, compressed);
}

void mxa_plink2compressed_shard(char *plink, char *plink_transposed, int snps_total, int indiv, int snp_begin, int snp_end, double *f,
                                int max_n, void **compressed) {
  clear_error();
  if (compressed) *compressed = nullptr;
  if (snp_begin < 0 || snp_end > snps_total || snp_begin >= snp_end || (snp_begin & 3)) {
    set_error(1, "mxa_plink2compressed_shard: need 0 <= snp_begin < snp_end <= snps_total and snp_begin %% 4 == 0");
    return;
  }
  const size_t ps = ((size_t)indiv + 3) / 4, pi = ((size_t)snps_total + 3) / 4;
  const uint8_t *p = reinterpret_cast<const uint8_t *>(plink) + (size_t)snp_begin * ps;
  const uint8_t *pt = (!plink_transposed || plink_transposed == plink) ? nullptr : reinterpret_cast<const uint8_t *>(plink_transposed) + (size_t)snp_begin / 4;
  (void)create_handle(p, ps, pt, pi, snp_end - snp_begin, indiv, f ? f + snp_begin : nullptr, max_n, compressed);
}

int mxa_plink2compressed_begin(long snps, long indiv, int max_n, void **compressed) {
  clear_error();
  return begin_handle(snps, indiv, max_n, compressed);
}
int mxa_plink2compressed_rows(void *compressed, const unsigned char *plink_rows, long snp_begin, long nrows, const double *f_rows) {
  clear_error();
  if (is_multi(compressed)) { set_error(16, "mxa_plink2compressed_rows: not available on a multi-device object"); return 1; }
  Handle *h = as_handle(compressed, "mxa_plink2compressed_rows");
  return h ? append_rows(h, plink_rows, snp_begin, nrows, f_rows) : 1;
}
int mxa_plink2compressed_end(void *compressed) {
  clear_error();
  if (is_multi(compressed)) { set_error(16, "mxa_plink2compressed_end: not available on a multi-device object"); return 1; }
  Handle *h = as_handle(compressed, "mxa_plink2compressed_end");
  return h ? end_handle(h) : 1;
}

static int trans_flag(const char *trans) {  // 5codesAPI.c:73-77
  if (!trans) exit(99);   // the reference dereferences it; a NULL letter is "anything else"
  if (*trans == 'T' || *trans == 't' || *trans == 'Y' || *trans == 'y') return 1;
  if (*trans != 'N' && *trans != 'n') exit(99);
  return 0;
}

void dgemm_compressed(char *trans, void *compressed, int n, double *B, int Ldb, double *C, int Ldc) {
  clear_error();
  const int t = trans_flag(trans);
  if (is_multi(compressed)) { (void)multi_gemm(compressed, t != 0, n, B, Ldb, C, Ldc); return; }
  Handle *h = as_handle(compressed, "dgemm_compressed");
  if (!h) return;
  (void)gemm_any(h, t != 0, n, B, Ldb, C, Ldc, Ldc, true, true);
}

int mxa_dgemm_compressed_device(char trans, void *compressed, int n, const double *dB, long ldb, double *dC, long ldc, void *hip_stream,
                                int sync) {
  clear_error();
  const int t = trans_flag(&trans);
  if (is_multi(compressed)) { set_error(16, "mxa_dgemm_compressed_device: not available on a multi-device object (MIRACULIX_NUM_GPUS > 1); use dgemm_compressed or mxa_dgemm_compressed_multi"); return 1; }
  Handle *h = as_handle(compressed, "mxa_dgemm_compressed_device");
  if (!h) return 1;
  if (n <= 0) return 0;
  if (refuse_while_staging(h, "mxa_dgemm_compressed_device")) return 1;
  MXA_HIP(hipSetDevice(h->device));
  hipStream_t s = hip_stream ? reinterpret_cast<hipStream_t>(hip_stream) : h->stream;
  if (gemm_device(h, t != 0, n, dB, ldb, dC, ldc, ldc, s, sync != 0)) return 1;
  if (sync) { MXA_HIP(hipStreamSynchronize(s)); harvest_profile(h); }
  return 0;
}

int mxa_gram_matvec(void *compressed, int n, const double *V, long ldv, double *out, long ldo) {
  clear_error();
  if (is_multi(compressed)) return multi_gram(compressed, n, V, ldv, out, ldo);
  Handle *h = as_handle(compressed, "mxa_gram_matvec");
  if (!h) return 1;
  return gram_any(h, n, V, ldv, out, ldo, true);
}

int mxa_gram_matvec_device(void *compressed, int n, const double *dV, long ldv, double *dOut, long ldo, int sync) {
  clear_error();
  if (is_multi(compressed)) { set_error(16, "mxa_gram_matvec_device: not available on a multi-device object; use mxa_gram_matvec"); return 1; }
  Handle *h = as_handle(compressed, "mxa_gram_matvec_device");
  if (!h) return 1;
  int vd = -1, od = -1;
  if (ptr_location(dV, &vd) != 1 || ptr_location(dOut, &od) != 1 || vd != h->device || od != h->device) {
    set_error(1, "mxa_gram_matvec_device: V and out must be memory of the object's device (%d)", h->device);
    return 1;
  }
  return gram_any(h, n, dV, ldv, dOut, ldo, sync != 0);
}

void free_compressed(void **compressed) {
  if (!compressed || !*compressed) return;
  if (is_multi(*compressed)) { multi_destroy(*compressed); *compressed = nullptr; return; }
  Handle *h = as_handle(*compressed, "free_compressed");
  if (h) destroy_handle(h);
  *compressed = nullptr;
}

// ------------------------------------------------------------------------------------------------ sparse x packed
namespace {
struct DevBuf {   // frees on scope exit
  void *p = nullptr;
  ~DevBuf() { if (p) (void)hipFree(p); }
  int alloc(size_t bytes) { MXA_HIP(hipMalloc(&p, bytes ? bytes : 1)); return 0; }
};
}  // namespace

static int sparse_times_plink_impl(bool tcompressed, const uint8_t *plink, const uint8_t *plink_transposed, int snps, int indiv, int nIdx,
                                   const int *rowIdxB, const int *colIdxB, const double *B, double *C, long ldc) {
  // reference semantics, pinned by running its library (tests/golden/make_golden_sparse.py): the sparse column index selects a ROW of
  // the packed matrix, the result runs over the 2-bit entries of that row:
  //   'N': P = plink (snps rows of indiv entries):             C (nIdx x indiv) = S (nIdx x snps)  * Z^T
  //   'T': P = plink_transposed (indiv rows of snps entries):  C (nIdx x snps)  = S (nIdx x indiv) * Z
  // zero-based CSR, uncentred, missing -> 0, C zero-filled over Ldc x entries (haplogeno.cc:1696).
  const uint8_t *P = tcompressed ? plink_transposed : plink;
  const long rows = tcompressed ? indiv : snps, entries = tcompressed ? snps : indiv;
  if (!P || !rowIdxB || !C || snps <= 0 || indiv <= 0 || nIdx < 0) { set_error(1, "sparse_times_plink: invalid argument"); return 1; }
  if (nIdx == 0) return 0;
  if (ldc < nIdx) { set_error(7, "sparse_times_plink: Ldc %ld < nIdx %d", ldc, nIdx); return 1; }
  const int dev = select_device();
  if (dev < 0) return 1;
  const size_t pitch = ((size_t)entries + 3) / 4;
  // CSR arrays on the host for validation
  std::vector<int> h_row((size_t)nIdx + 1);
  MXA_HIP(hipMemcpy(h_row.data(), rowIdxB, sizeof(int) * ((size_t)nIdx + 1), is_device_ptr(rowIdxB) ? hipMemcpyDeviceToHost : hipMemcpyHostToHost));
  const long nnz = h_row[nIdx];
  if (h_row[0] != 0 || nnz < 0) { set_error(1, "sparse_times_plink: rowIdxB must be zero-based CSR row pointers (rowIdxB[0] = %d)", h_row[0]); return 1; }
  for (int j = 0; j < nIdx; j++) if (h_row[j + 1] < h_row[j]) { set_error(1, "sparse_times_plink: rowIdxB is not non-decreasing at row %d", j); return 1; }
  if (nnz > 0 && (!colIdxB || !B)) { set_error(1, "sparse_times_plink: colIdxB / B are NULL"); return 1; }
  std::vector<int> h_col((size_t)nnz);
  if (nnz) MXA_HIP(hipMemcpy(h_col.data(), colIdxB, sizeof(int) * (size_t)nnz, is_device_ptr(colIdxB) ? hipMemcpyDeviceToHost : hipMemcpyHostToHost));
  for (long t = 0; t < nnz; t++)
    if (h_col[t] < 0 || h_col[t] >= rows) { set_error(1, "sparse_times_plink: colIdxB[%ld] = %d outside [0, %ld)", t, h_col[t], rows); return 1; }

  // packed rows on the device: a device matrix is used in place; from the host only the rows S refers to are uploaded
  DevBuf dP;
  const uint8_t *d_packed = P;
  if (!is_device_ptr(P)) {
    std::vector<int> remap((size_t)rows, -1), used;
    for (long t = 0; t < nnz; t++) if (remap[h_col[t]] < 0) { remap[h_col[t]] = 0; used.push_back(h_col[t]); }
    std::sort(used.begin(), used.end());
    for (size_t u = 0; u < used.size(); u++) remap[used[u]] = (int)u;
    for (long t = 0; t < nnz; t++) h_col[t] = remap[h_col[t]];
    if (dP.alloc(used.size() * pitch)) return 1;
    const size_t rows_per_chunk = std::max<size_t>(1, ((size_t)256 << 20) / pitch);
    std::vector<uint8_t> bounce(std::min(rows_per_chunk, std::max<size_t>(1, used.size())) * pitch);
    for (size_t u0 = 0; u0 < used.size(); u0 += rows_per_chunk) {
      const size_t cnt = std::min(rows_per_chunk, used.size() - u0);
      for (size_t u = 0; u < cnt; u++) memcpy(bounce.data() + u * pitch, P + (size_t)used[u0 + u] * pitch, pitch);
      MXA_HIP(hipMemcpy(static_cast<uint8_t *>(dP.p) + u0 * pitch, bounce.data(), cnt * pitch, hipMemcpyHostToDevice));
    }
    d_packed = static_cast<const uint8_t *>(dP.p);
  }
  DevBuf dRow, dCol, dVal;
  if (dRow.alloc(sizeof(int) * ((size_t)nIdx + 1)) || dCol.alloc(sizeof(int) * (size_t)nnz) || dVal.alloc(sizeof(double) * (size_t)nnz)) return 1;
  MXA_HIP(hipMemcpy(dRow.p, h_row.data(), sizeof(int) * ((size_t)nIdx + 1), hipMemcpyHostToDevice));
  if (nnz) {
    MXA_HIP(hipMemcpy(dCol.p, h_col.data(), sizeof(int) * (size_t)nnz, hipMemcpyHostToDevice));
    MXA_HIP(hipMemcpy(dVal.p, B, sizeof(double) * (size_t)nnz, is_device_ptr(B) ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice));
  }
  hipStream_t s = nullptr;   // legacy default stream: ordered with everything else of the process
  if (is_device_ptr(C)) {
    if (launch_sparse_times_plink(d_packed, pitch, entries, nIdx, (const int *)dRow.p, (const int *)dCol.p, (const double *)dVal.p, C, ldc, 0, entries, s)) return 1;
    MXA_HIP(hipStreamSynchronize(s));
    return 0;
  }
  // host C: slabs of columns through a device buffer of at most 1 GiB
  long e_slab = std::max<long>(512, (((long)1 << 30) / (8 * ldc)) / 512 * 512);
  e_slab = std::min(e_slab, (entries + 511) / 512 * 512);
  DevBuf dC;
  if (dC.alloc(sizeof(double) * (size_t)ldc * (size_t)e_slab)) return 1;
  for (long e0 = 0; e0 < entries; e0 += e_slab) {
    const long cnt = std::min(e_slab, entries - e0);
    if (launch_sparse_times_plink(d_packed, pitch, entries, nIdx, (const int *)dRow.p, (const int *)dCol.p, (const double *)dVal.p, (double *)dC.p, ldc, e0, cnt, s)) return 1;
    MXA_HIP(hipMemcpy(C + (size_t)e0 * ldc, dC.p, sizeof(double) * (size_t)ldc * (size_t)cnt, hipMemcpyDeviceToHost));
  }
  return 0;
}

void sparse_times_plink(char *transsparse, char *transcompressed, char *plink, char *plink_transposed, int snps, int indiv, int nIdx, int *rowIdxB,
                        int *colIdxB, double *B, double *C, int Ldc) {
  clear_error();
  const int tc = trans_flag(transcompressed);
  if (trans_flag(transsparse)) {   // reference: BUG -> "Severe error occured in function 'sparseTGeno'" + exit (haplogeno.cc:1698)
    fprintf(stderr, "miraculix_amd: sparse_times_plink: a transposed sparse matrix is not supported (nor by the reference)\n");
    exit(EXIT_FAILURE);
  }
  (void)sparse_times_plink_impl(tc != 0, reinterpret_cast<const uint8_t *>(plink), reinterpret_cast<const uint8_t *>(plink_transposed), snps, indiv, nIdx,
                                rowIdxB, colIdxB, B, C, Ldc);
}

// dgemm_plink (5codesAPI.c:112-130): the documented semantics as plink2compressed + dgemm_compressed + free_compressed; see the header (parity unpinned:
// the reference aborts unconditionally there).
void dgemm_plink(char *trans, char *plink, char *plink_transposed, int snps, int indiv, double *f, int n, double *B, int Ldb, double *C, int Ldc) {
  clear_error();
  const int t = trans_flag(trans);
  if (snps <= 0 || indiv <= 0 || n < 0) { set_error(1, "dgemm_plink: snps and indiv must be positive"); return; }
  if (n == 0) return;
  if (!plink && !plink_transposed) { set_error(1, "dgemm_plink: plink and plink_transposed are both NULL"); return; }
  if (select_device() < 0) return;
  const size_t bps = ((size_t)indiv + 3) / 4, bpi = ((size_t)snps + 3) / 4;
  DevBuf tmp;
  const uint8_t *p_snp = reinterpret_cast<const uint8_t *>(plink);
  if (!p_snp) {   // only the individual-major matrix was handed over ('N' in the reference's call shape): the SNP-major copy is made on the device
    DevBuf src;
    const uint8_t *pt = reinterpret_cast<const uint8_t *>(plink_transposed);
    if (!is_device_ptr(pt)) {
      if (src.alloc((size_t)indiv * bpi) || !check_hip(hipMemcpy(src.p, pt, (size_t)indiv * bpi, hipMemcpyHostToDevice), __func__, __LINE__)) return;
      pt = static_cast<const uint8_t *>(src.p);
    }
    if (tmp.alloc((size_t)snps * bps) || launch_transpose_2bit(pt, indiv, snps, static_cast<uint8_t *>(tmp.p), nullptr) ||
        !check_hip(hipDeviceSynchronize(), __func__, __LINE__)) return;
    p_snp = static_cast<const uint8_t *>(tmp.p);
  }
  Options &o = options();
  const Options saved = o;
  o.gpu = true; o.centered = f != nullptr; o.set = true;
  void *obj = nullptr;
  tl_single_override = 1;                 // one product: the SNP-major copy serves it whichever way
  tl_no_warmup = true;
  const int rc = create_handle(p_snp, bps, nullptr, bpi, snps, indiv, f, n, &obj);
  tl_single_override = -1; tl_no_warmup = false;
  if (!rc && obj) {
    Handle *h = reinterpret_cast<Handle *>(obj);
    (void)gemm_any(h, t != 0, n, B, Ldb, C, Ldc, Ldc, true, true);
    destroy_handle(h);
  }
  o = saved;
}

void get_compressed_freq(void *compressed, double *f) {
  clear_error();
  if (is_multi(compressed)) { if (f) multi_freq(compressed, f); return; }
  Handle *h = as_handle(compressed, "get_compressed_freq");
  if (!h || !f) return;
  if (refuse_while_staging(h, "get_compressed_freq")) return;   // the host copy of the frequencies is filled when the object is sealed
  memcpy(f, h->h_f, sizeof(double) * (size_t)h->snps);
}

int mxa_last_error(void) { return g_err; }
const char *mxa_last_error_string(void) { return g_errmsg; }

int mxa_device_count(void) {
  int c = 0;
  if (hipGetDeviceCount(&c) != hipSuccess) { (void)hipGetLastError(); return -1; }
  return c;
}

int mxa_set_engine(int engine) {
  if (engine != 0 && engine != 1 && engine != 3 && engine != 4) return g_engine.load();
  return g_engine.exchange(engine);
}
int mxa_get_engine(void) { return g_engine.load(); }

void mxa_profile_reset(void) { profile() = Profile(); }
void mxa_profile_get(int *launches, double *total_ms) {
  if (launches) *launches = profile().launches;
  if (total_ms) *total_ms = profile().total_ms;
}
int mxa_last_path(void) {
  Geometry g;
  { std::lock_guard<std::mutex> lk(g_prof_mutex); g = last_geometry(); }
  if (g.path != 4) return g.path;
  // n <= 2 under the default engine: the route was chosen on the device; read the verdict now (the entries that use it are synchronous)
  int flag = 0, prev = 0;
  if (!g.d_flag) return 2;
  (void)hipGetDevice(&prev);
  if (hipSetDevice(g.flag_dev) != hipSuccess || hipMemcpy(&flag, g.d_flag, sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) { (void)hipGetLastError(); flag = 0; }
  (void)hipSetDevice(prev);
  return flag ? 3 : 2;
}
int mxa_last_range_fallback(void *compressed) {
  if (!compressed || is_multi(compressed)) return -1;
  Handle *h = as_handle(compressed, "mxa_last_range_fallback");
  if (!h || !h->ws.d_denflag) return -1;
  int flag = 0, prev = 0;
  (void)hipGetDevice(&prev);
  if (hipSetDevice(h->device) != hipSuccess || hipStreamSynchronize(h->stream) != hipSuccess ||
      hipMemcpy(&flag, h->ws.d_denflag, sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) { (void)hipGetLastError(); flag = -1; }
  (void)hipSetDevice(prev);
  return flag;
}
void mxa_last_geometry(long *m, long *k, int *n, int *splits, int *a_tile, int *c_tile) {
  const Geometry &g = last_geometry();
  if (m) *m = g.m; if (k) *k = g.k; if (n) *n = g.n; if (splits) *splits = g.splits; if (a_tile) *a_tile = g.a; if (c_tile) *c_tile = g.c;
}

long mxa_plan_partial_doubles(long m, long k, int n) {
  if (m <= 0 || k <= 0 || n <= 0) return 0;
  const GemmPlan p = plan_gemm(m, (k + kSlabK - 1) / kSlabK * kSlabK, n);
  return (long)p.splits * p.n_pad * p.m_pad;
}
int mxa_single_orientation(void *compressed) {
  if (!compressed) return -1;
  if (is_multi(compressed)) return multi_single(compressed);
  Handle *h = as_handle(compressed, "mxa_single_orientation");
  return h ? (h->single ? 1 : 0) : -1;
}
long mxa_partial_capacity(void *compressed) {
  if (!compressed || is_multi(compressed)) return -1;
  Handle *h = as_handle(compressed, "mxa_partial_capacity");
  return h ? (long)h->ws.cap_P : -1;
}

// ---- .bed staging owned by the library
static long count_lines(const char *path) {
  FILE *fh = fopen(path, "rb");
  if (!fh) return -1;
  long n = 0; int c, last = '\n';
  while ((c = fgetc(fh)) != EOF) { if (c == '\n') n++; last = c; }
  if (last != '\n') n++;
  fclose(fh);
  return n;
}

static std::string bed_base(const char *bed_path) {
  std::string base(bed_path);
  if (base.size() > 4 && base.compare(base.size() - 4, 4, ".bed") == 0) base.resize(base.size() - 4);
  return base;
}

}  // extern "C"

namespace mxa {
// SNP rows [snp_begin, snp_end) of the SNP-major file `base`.bed -> object on `device`.  Only those rows are read (the reference
// reader streams the whole file row by row, read_plink.jl:161-222); the individual-major block and the allele frequencies of the
// range are produced on the device (k_transpose_2bit*, k_allele_freq), so no process ever needs the full individual-major matrix.
// File chunks go through two pinned buffers so that reading chunk c+1 overlaps the upload of chunk c.
int bed_range_to_handle(const char *base_c, long snps_total, long indiv, long snp_begin, long snp_end, int max_n, int device, void **out,
                        double *f_out_local) {
  if (out) *out = nullptr;
  const std::string base(base_c);
  if (!out || snps_total <= 0 || indiv <= 0 || snp_begin < 0 || snp_end > snps_total || snp_begin >= snp_end) {
    set_error(1, "bed staging: need 0 <= snp_begin < snp_end <= snps (got [%ld, %ld) of %ld) and indiv > 0", snp_begin, snp_end, snps_total);
    return 1;
  }
  FILE *fh = fopen((base + ".bed").c_str(), "rb");
  if (!fh) { set_error(1, "mxa_bed2compressed: cannot open %s.bed", base.c_str()); return 1; }
  unsigned char magic[3];
  if (fread(magic, 1, 3, fh) != 3 || magic[0] != 0x6c || magic[1] != 0x1b || magic[2] != 0x01) {
    fclose(fh);
    set_error(1, "mxa_bed2compressed: %s.bed is not a SNP-major PLINK .bed file (magic bytes 6c 1b 01 expected)", base.c_str());
    return 1;
  }
  const size_t bps = ((size_t)indiv + 3) / 4;
  const long rows = snp_end - snp_begin;
  const size_t bpi = ((size_t)rows + 3) / 4;
  // the reference reader fails on a file that is too short and asserts eof afterwards ("Too large .bed file", read_plink.jl:190-193)
  if (fseeko(fh, 0, SEEK_END) == 0) {
    const off_t sz = ftello(fh);
    if (sz >= 0 && (size_t)sz != 3 + (size_t)snps_total * bps) {
      fclose(fh);
      set_error(1, "mxa_bed2compressed: %s.bed has %lld bytes, expected 3 + %ld x ceil(%ld/4) = %zu", base.c_str(), (long long)sz, snps_total, indiv, 3 + (size_t)snps_total * bps);
      return 1;
    }
  }
  if (fseeko(fh, (off_t)(3 + (size_t)snp_begin * bps), SEEK_SET) != 0) { fclose(fh); set_error(1, "mxa_bed2compressed: cannot seek in %s.bed", base.c_str()); return 1; }
  int dev = device;
  if (dev < 0) dev = select_device();
  else if (!check_hip(hipSetDevice(dev), __func__, __LINE__)) dev = -1;
  if (dev < 0) { fclose(fh); return 1; }
  uint8_t *d_plink = nullptr, *d_plink_t = nullptr;
  double *d_f = nullptr;
  void *pin[2] = {nullptr, nullptr};
  hipEvent_t ev[2] = {nullptr, nullptr};
  hipStream_t cs = nullptr;
  int rc = 0;
  auto bad = [&](hipError_t e, int line) { if (e != hipSuccess) { check_hip(e, "mxa_bed2compressed", line); rc = 1; } return rc; };
  const size_t rows_per_chunk = std::max<size_t>(1, ((size_t)64 << 20) / bps);
  const size_t chunk_bytes = std::min(rows_per_chunk, (size_t)rows) * bps;
  // only the SNP-major copy will be kept (no transposed block is built): asked for, or the raw block + its transpose + the two packed copies do not fit
  bool single;
  {
    const int policy = tl_single_override >= 0 ? tl_single_override : single_orientation_policy();
    size_t free_b = 0, total_b = 0;
    if (bad(hipMemGetInfo(&free_b, &total_b), __LINE__)) goto out;
    const size_t raw = (size_t)rows * bps, raw_t = (size_t)indiv * bpi;
    single = policy == 1 || (policy == 2 && raw + raw_t + object_footprint(rows, indiv, max_n, false) > free_b && 2 * chunk_bytes + object_footprint(rows, indiv, max_n, true) <= free_b);
  }
  if (single) {
    // ONE packed copy: the file is streamed straight into the object (round 5) -- chunk c + 1 is read from the file while chunk c is uploaded, recoded into
    // the tiled layout and counted (k_recode, k_allele_freq), all on the object's stream; device memory beyond the object: two 64 MB chunks.  (Round 4 held
    // the raw matrix on the device beside the object: a 250 GB .bed could not become a 250 GB object.)
    uint8_t *dchunk[2] = {nullptr, nullptr};
    Handle *h = nullptr;
    {
      void *obj = nullptr;
      if (begin_handle(rows, indiv, max_n, &obj, dev)) { rc = 1; goto out; }
      h = reinterpret_cast<Handle *>(obj);
      *out = obj;
    }
    for (int i = 0; i < 2 && !rc; i++)
      if (bad(hipMalloc((void **)&dchunk[i], chunk_bytes), __LINE__) || bad(hipHostMalloc(&pin[i], chunk_bytes, hipHostMallocDefault), __LINE__) ||
          bad(hipEventCreateWithFlags(&ev[i], hipEventDisableTiming), __LINE__)) break;
    {
      int slot = 0;
      bool used[2] = {false, false};
      for (size_t r0 = 0; r0 < (size_t)rows && !rc; r0 += rows_per_chunk, slot ^= 1) {
        const size_t nr = std::min(rows_per_chunk, (size_t)rows - r0);
        if (used[slot] && bad(hipEventSynchronize(ev[slot]), __LINE__)) break;     // upload and kernels that last used this pair of buffers are done
        if (fread(pin[slot], 1, nr * bps, fh) != nr * bps) { set_error(1, "mxa_bed2compressed: %s.bed is shorter than %ld x ceil(%ld/4) bytes", base.c_str(), snps_total, indiv); rc = 1; break; }
        if (bad(hipMemcpyAsync(dchunk[slot], pin[slot], nr * bps, hipMemcpyHostToDevice, h->stream), __LINE__)) break;
        if (launch_recode(dchunk[slot], bps, (long)r0, (long)nr, indiv, 0, h->snp_major, h->stream) ||
            launch_allele_freq(dchunk[slot], (long)nr, indiv, h->d_f + r0, h->stream)) { rc = 1; break; }
        if (bad(hipEventRecord(ev[slot], h->stream), __LINE__)) break;
        used[slot] = true;
      }
      if (!rc) bad(hipStreamSynchronize(h->stream), __LINE__);
    }
    if (!rc) { h->staged_rows = rows; h->staged_iv.assign(1, {0L, (long)rows}); rc = end_handle(h); }   // (the loop above wrote every row of the range exactly once)
    if (!rc && f_out_local) memcpy(f_out_local, h->h_f, sizeof(double) * rows);
    for (int i = 0; i < 2; i++) if (dchunk[i]) (void)hipFree(dchunk[i]);
    goto out;
  }
  if (bad(hipMalloc((void **)&d_plink, (size_t)rows * bps), __LINE__) || (!single && bad(hipMalloc((void **)&d_plink_t, (size_t)indiv * bpi), __LINE__)) ||
      bad(hipMalloc((void **)&d_f, sizeof(double) * rows), __LINE__) || bad(hipStreamCreateWithFlags(&cs, hipStreamNonBlocking), __LINE__)) goto out;
  for (int i = 0; i < 2; i++)
    if (bad(hipHostMalloc(&pin[i], chunk_bytes, hipHostMallocDefault), __LINE__) || bad(hipEventCreateWithFlags(&ev[i], hipEventDisableTiming), __LINE__)) goto out;
  {
    int slot = 0;
    bool used[2] = {false, false};
    for (size_t r0 = 0; r0 < (size_t)rows && !rc; r0 += rows_per_chunk, slot ^= 1) {
      const size_t nr = std::min(rows_per_chunk, (size_t)rows - r0);
      if (used[slot] && bad(hipEventSynchronize(ev[slot]), __LINE__)) break;     // the upload that last used this buffer is done
      if (fread(pin[slot], 1, nr * bps, fh) != nr * bps) { set_error(1, "mxa_bed2compressed: %s.bed is shorter than %ld x ceil(%ld/4) bytes", base.c_str(), snps_total, indiv); rc = 1; break; }
      if (bad(hipMemcpyAsync(d_plink + r0 * bps, pin[slot], nr * bps, hipMemcpyHostToDevice, cs), __LINE__) || bad(hipEventRecord(ev[slot], cs), __LINE__)) break;
      used[slot] = true;
    }
    if (!rc) bad(hipStreamSynchronize(cs), __LINE__);
  }
  if (!rc && !single) rc = launch_transpose_2bit(d_plink, rows, indiv, d_plink_t, nullptr);
  if (!rc) rc = launch_allele_freq(d_plink, rows, indiv, d_f, nullptr);
  if (!rc) bad(hipDeviceSynchronize(), __LINE__);
  if (!rc) {
    const int keep = tl_single_override;
    tl_single_override = single ? 1 : 0;     // the object follows the decision taken for the staging buffers
    rc = create_handle(d_plink, bps, d_plink_t, bpi, rows, indiv, d_f, max_n, out, dev);
    tl_single_override = keep;
  }
  if (!rc && f_out_local) bad(hipMemcpy(f_out_local, d_f, sizeof(double) * rows, hipMemcpyDeviceToHost), __LINE__);
out:
  fclose(fh);
  for (int i = 0; i < 2; i++) { if (pin[i]) (void)hipHostFree(pin[i]); if (ev[i]) (void)hipEventDestroy(ev[i]); }
  if (cs) (void)hipStreamDestroy(cs);
  if (d_plink) (void)hipFree(d_plink);
  if (d_plink_t) (void)hipFree(d_plink_t);
  if (d_f) (void)hipFree(d_f);
  if (rc && out && *out) { destroy_handle(reinterpret_cast<Handle *>(*out)); *out = nullptr; }
  return rc;
}
}  // namespace mxa

extern "C" {

static int bed_dims(const char *who, const std::string &base, int *snps, int *indiv) {
  if (*snps <= 0) *snps = (int)count_lines((base + ".bim").c_str());
  if (*indiv <= 0) *indiv = (int)count_lines((base + ".fam").c_str());
  if (*snps <= 0 || *indiv <= 0) { set_error(1, "%s: dimensions unknown (no .bim/.fam next to %s.bed)", who, base.c_str()); return 1; }
  return 0;
}

int mxa_bed2compressed(const char *bed_path, int snps, int indiv, int max_n, void **compressed, double *f_out, int *snps_out, int *indiv_out) {
  clear_error();
  if (compressed) *compressed = nullptr;
  if (!bed_path || !compressed) { set_error(1, "mxa_bed2compressed: bad arguments"); return 1; }
  const std::string base = bed_base(bed_path);
  if (bed_dims("mxa_bed2compressed", base, &snps, &indiv)) return 1;
  int rc;
  const int shards = multi_requested();
  if (shards > 1 || getenv("MXA_FORCE_MULTI")) rc = multi_create_from_bed(base.c_str(), snps, indiv, max_n, shards, compressed, f_out);
  else rc = bed_range_to_handle(base.c_str(), snps, indiv, 0, snps, max_n, -1, compressed, f_out);
  if (!rc) { if (snps_out) *snps_out = snps; if (indiv_out) *indiv_out = indiv; }
  return rc;
}

int mxa_bed2compressed_range(const char *bed_path, int snps, int indiv, int snp_begin, int snp_end, int max_n, void **compressed, double *f_out) {
  clear_error();
  if (compressed) *compressed = nullptr;
  if (!bed_path || !compressed) { set_error(1, "mxa_bed2compressed_range: bad arguments"); return 1; }
  const std::string base = bed_base(bed_path);
  if (bed_dims("mxa_bed2compressed_range", base, &snps, &indiv)) return 1;
  return bed_range_to_handle(base.c_str(), snps, indiv, snp_begin, snp_end, max_n, -1, compressed, f_out);
}

// ---- staging helpers
int mxa_transpose_2bit(const unsigned char *in, long rows, long cols, unsigned char *out) {
  clear_error();
  if (!in || !out || rows <= 0 || cols <= 0) { set_error(1, "mxa_transpose_2bit: bad arguments"); return 1; }
  if (select_device() < 0) return 1;
  const size_t nin = (size_t)rows * ((cols + 3) / 4), nout = (size_t)cols * ((rows + 3) / 4);
  const bool in_dev = is_device_ptr(in), out_dev = is_device_ptr(out);
  uint8_t *d_in = const_cast<uint8_t *>(in), *d_out = out;
  int rc = 0;
  if (!in_dev) { MXA_HIP(hipMalloc((void **)&d_in, nin)); if (!check_hip(hipMemcpy(d_in, in, nin, hipMemcpyHostToDevice), __func__, __LINE__)) rc = 1; }
  if (!out_dev && !rc) { if (!check_hip(hipMalloc((void **)&d_out, nout), __func__, __LINE__)) rc = 1; }
  if (!rc) rc = launch_transpose_2bit(d_in, rows, cols, d_out, nullptr);
  if (!rc && !check_hip(hipDeviceSynchronize(), __func__, __LINE__)) rc = 1;
  if (!rc && !out_dev && !check_hip(hipMemcpy(out, d_out, nout, hipMemcpyDeviceToHost), __func__, __LINE__)) rc = 1;
  if (!in_dev && d_in) (void)hipFree(d_in);
  if (!out_dev && d_out) (void)hipFree(d_out);
  return rc;
}

int mxa_allele_freq(const unsigned char *plink, long snps, long indiv, double *f) {
  clear_error();
  if (!plink || !f || snps <= 0 || indiv <= 0) { set_error(1, "mxa_allele_freq: bad arguments"); return 1; }
  if (select_device() < 0) return 1;
  const size_t nin = (size_t)snps * ((indiv + 3) / 4);
  const bool in_dev = is_device_ptr(plink), out_dev = is_device_ptr(f);
  uint8_t *d_in = const_cast<uint8_t *>(plink);
  double *d_f = f;
  int rc = 0;
  if (!in_dev) { MXA_HIP(hipMalloc((void **)&d_in, nin)); if (!check_hip(hipMemcpy(d_in, plink, nin, hipMemcpyHostToDevice), __func__, __LINE__)) rc = 1; }
  if (!out_dev && !rc) { if (!check_hip(hipMalloc((void **)&d_f, sizeof(double) * snps), __func__, __LINE__)) rc = 1; }
  if (!rc) rc = launch_allele_freq(d_in, snps, indiv, d_f, nullptr);
  if (!rc && !check_hip(hipDeviceSynchronize(), __func__, __LINE__)) rc = 1;
  if (!rc && !out_dev && !check_hip(hipMemcpy(f, d_f, sizeof(double) * snps, hipMemcpyDeviceToHost), __func__, __LINE__)) rc = 1;
  if (!in_dev && d_in) (void)hipFree(d_in);
  if (!out_dev && d_f) (void)hipFree(d_f);
  return rc;
}

}  // extern "C"
