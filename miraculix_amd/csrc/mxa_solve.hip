// mxa_solve.hip -- the solver twin of the reference (src/cuda/solve_cuda.cu:70-951; Julia binding src/bindings/Julia/solve.jl,
// Fortran binding src/bindings/Fortran/modmiraculix_gpu.f90): dense Cholesky solve + log-determinant and sparse triangular solves.
// SURVEY.md 8(f)-4, last item: not part of the compressed-genotype hot path.
//
// The reference calls cuSOLVER Xpotrf/Xpotrs and cuSPARSE SpSM.  Their ROCm counterparts cannot be used the same way here:
// measured on MI355X / ROCm 7.2, dlopen() of librocsolver.so or librocsparse.so in a process whose HIP runtime is already
// initialised (the normal case: plink2compressed has run) takes 130 s to > 4 min (their thousands of code objects are loaded
// eagerly), and even librocblas.so takes 2.4 s.  Round 3: no vendor library at all -- everything is written here / in mxa_dense.hip:
//   dense   two-level blocked right-looking Cholesky: k_potrf_block (64 x 64 diagonal block in LDS), k_trtri_blocks (its inverse), and
//           k_dgemm (fp64 MFMA) for the block column below it (A L^-T as a product with the inverted block), for the rest of the 512-column
//           panel, and once per panel -- lower tiles only -- for the trailing matrix; the two triangular solves block by block as
//           products with the inverted diagonal blocks; k_logdet (the reference's trace_kernel, :884-909)
//   sparse  k_sptrsm: synchronisation-free triangular solve, one wave per row, rows in dependency order, a flag per row
//           (host side: COO -> sorted CSR of A and of A^T, diagonal check)
#include "../../include/miraculix_amd.h"
#include "mxa_internal.h"


#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <numeric>
#include <string>
#include <vector>

namespace mxa {

static bool solve_is_device_ptr(const void *p) {
  if (!p) return false;
  hipPointerAttribute_t attr;
  if (hipPointerGetAttributes(&attr, p) != hipSuccess) { (void)hipGetLastError(); return false; }
  return attr.type == hipMemoryTypeDevice || attr.type == hipMemoryTypeManaged;
}
static int solve_select_device(const char *who) {
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) { (void)hipGetLastError(); set_error(10, "%s: no HIP device available; this engine is GPU-only", who); return 1; }
  const char *dv = getenv("HIP_DEVICE");
  if (!dv) dv = getenv("CUDA_DEVICE");
  if (dv) MXA_HIP(hipSetDevice(atoi(dv)));
  return 0;
}

// ------------------------------------------------------------------------------------------------ dense: Cholesky solve + logdet
constexpr int kPotrfNB = 64;        // diagonal block factored in LDS
constexpr int kPotrfPanel = 512;    // columns per outer panel

// logdet(A) = sum_i 2 log L_ii of the Cholesky factor (reference trace_kernel, solve_cuda.cu:884-909, which adds with atomics);
// here one workgroup, fixed-order tree: bitwise reproducible
__global__ void __launch_bounds__(1024) k_logdet(const double *__restrict__ L, long n, double *__restrict__ out) {
  __shared__ double sh[1024];
  double s = 0.0;
  for (long i = threadIdx.x; i < n; i += 1024) s += 2.0 * log(L[i * (n + 1)]);
  sh[threadIdx.x] = s;
  __syncthreads();
  for (int w = 512; w > 0; w >>= 1) { if ((int)threadIdx.x < w) sh[threadIdx.x] += sh[threadIdx.x + w]; __syncthreads(); }
  if (threadIdx.x == 0) *out = sh[0];
}

namespace {
struct Dev {   // frees on scope exit
  void *p = nullptr;
  ~Dev() { if (p) (void)hipFree(p); }
};
}  // namespace

static int dense_solve_impl(const double *A, unsigned int input_size, const double *B, unsigned int rhs_cols, double *X, double *logdet,
                            int oversubscribe) {
  if (!A || !B || !X || input_size == 0 || rhs_cols == 0) { set_error(1, "potrs_solve_gpu: invalid argument"); return 1; }
  if (oversubscribe != 0 && oversubscribe != 1) { set_error(1, "potrs_solve_gpu: oversubscribe must be 0 or 1"); return 1; }
  if (solve_select_device("potrs_solve_gpu")) return 1;
  const size_t n = input_size, nrhs = rhs_cols;
  const long nblk = (long)((n + kPotrfNB - 1) / kPotrfNB);
  Dev dA, dB, dInfo, dLog, dInv;
  if (oversubscribe) MXA_HIP(hipMallocManaged(&dA.p, sizeof(double) * n * n));
  else MXA_HIP(hipMalloc(&dA.p, sizeof(double) * n * n));
  MXA_HIP(hipMalloc(&dB.p, sizeof(double) * n * nrhs));
  MXA_HIP(hipMalloc(&dInfo.p, sizeof(int)));
  MXA_HIP(hipMalloc(&dInv.p, sizeof(double) * kPotrfNB * kPotrfNB * (size_t)nblk));   // inverted diagonal blocks
  MXA_HIP(hipMemset(dInfo.p, 0, sizeof(int)));
  MXA_HIP(hipMemcpy(dA.p, A, sizeof(double) * n * n, solve_is_device_ptr(A) ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice));
  MXA_HIP(hipMemcpy(dB.p, B, sizeof(double) * n * nrhs, solve_is_device_ptr(B) ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice));
  const auto tick = [] { (void)hipDeviceSynchronize(); return std::chrono::steady_clock::now(); };
  const auto secs = [](std::chrono::steady_clock::time_point a0, std::chrono::steady_clock::time_point a1) { return std::chrono::duration<double>(a1 - a0).count(); };
  const bool verbose = env_print_level() > 0;   // timings like the reference's debug_info lines (solve_cuda.cu:157-259)
  auto t_start = verbose ? tick() : std::chrono::steady_clock::time_point();
  double *a = static_cast<double *>(dA.p), *b = static_cast<double *>(dB.p), *inv = static_cast<double *>(dInv.p);
  const long N = (long)n;
  hipStream_t st = nullptr;
  int rc = 0;
  do {
    // lower triangle of the column-major image, as the reference (CUBLAS_FILL_MODE_LOWER, solve_cuda.cu:84); A is symmetric.
    // Two-level right-looking Cholesky.  Outer panels of kPotrfPanel columns; inside a panel, per 64-column step: factor the
    // diagonal block, invert it, solve the block column below it (full height) as A21 <- A21 L11^-T = A21 (L11^-1)^T -- a product, in
    // place: a workgroup reads only the rows it writes --, update only the REST OF THE PANEL; after the panel one rank-512 update
    // of the lower tiles of the whole trailing matrix (an update per 64 columns re-reads it 8x as often and is memory-bound).
    // LOOKAHEAD (round 3): the in-panel steps are a chain of ~32 small dependent launches (~1 ms per panel, 29 ms of a 66 ms factorisation at
    // n = 15 000) during which the chip idles.  Panel p's update is therefore split: its part for the columns of panel p + 1 (L(p)) stays on the
    // factorisation stream, the rest (B(p): columns from panel p + 2 on) goes to a second stream, and panel p + 1 is factored while B(p) runs.
    // Ordering: B(p) after F(p) (event) and after B(p - 1) (same stream); L(p) after B(p - 1) (both update panel p + 1's columns: event);
    // F(p + 1) after L(p) (same stream).  The second stream then runs the big updates back to back.
    const size_t npanels = (n + kPotrfPanel - 1) / kPotrfPanel;
    struct Ev { std::vector<hipEvent_t> v; ~Ev() { for (hipEvent_t e : v) if (e) (void)hipEventDestroy(e); } } evF, evB;
    struct St { hipStream_t s = nullptr; ~St() { if (s) (void)hipStreamDestroy(s); } } sA, sB;
    evF.v.assign(npanels, nullptr); evB.v.assign(npanels, nullptr);
    // the factorisation chain gets the higher priority: its one-workgroup launches must not queue behind the thousands of workgroups of an update
    int prio_lo = 0, prio_hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
    if (!check_hip(hipStreamCreateWithPriority(&sA.s, hipStreamNonBlocking, prio_hi), __func__, __LINE__) ||
        !check_hip(hipStreamCreateWithPriority(&sB.s, hipStreamNonBlocking, prio_lo), __func__, __LINE__)) { rc = 1; break; }
    st = sA.s;
    for (size_t q = 0; q < npanels && !rc; q++)
      if (!check_hip(hipEventCreateWithFlags(&evF.v[q], hipEventDisableTiming), __func__, __LINE__) ||
          !check_hip(hipEventCreateWithFlags(&evB.v[q], hipEventDisableTiming), __func__, __LINE__)) rc = 1;
    if (rc) break;
    for (size_t K = 0, pi = 0; K < n && !rc; K += kPotrfPanel, pi++) {
      const size_t pw = std::min<size_t>(kPotrfPanel, n - K);          // panel width
      for (size_t k = K; k < K + pw && !rc; k += kPotrfNB) {           // F(pi)
        const int nb = (int)std::min<size_t>(kPotrfNB, K + pw - k);
        double *inv_k = inv + (k / kPotrfNB) * (size_t)(kPotrfNB * kPotrfNB);   // kept: the solves of the right-hand sides use it again
        if (launch_potrf_inv_block(a + k + k * n, N, nb, (long)k, static_cast<int *>(dInfo.p), inv_k, st)) { rc = 1; break; }
        const long below = (long)(n - k - nb);                         // rows under the diagonal block
        if (below <= 0) continue;
        double *a21 = a + (k + nb) + k * n;
        if (launch_dgemm(false, true, below, nb, nb, 1.0, a21, N, inv_k, kPotrfNB, 0.0, a21, N, false, st)) { rc = 1; break; }
        const long rest = (long)(K + pw - k - nb);                     // columns of the panel still to the right
        if (rest > 0 && launch_dgemm(false, true, below, rest, nb, -1.0, a21, N, a21, N, 1.0, a + (k + nb) + (k + nb) * n, N, false, st)) { rc = 1; break; }
      }
      if (rc) break;
      const long trailing = (long)(n - K - pw);
      if (trailing <= 0) break;
      if (!check_hip(hipEventRecord(evF.v[pi], st), __func__, __LINE__)) { rc = 1; break; }
      const long pw1 = std::min<long>(kPotrfPanel, trailing);          // width of the next panel
      const double *p21 = a + (K + pw) + K * n;                        // panel pi below its own diagonal blocks: trailing x pw
      // L(pi): the next panel's columns, all trailing rows
      if (pi > 0 && !check_hip(hipStreamWaitEvent(st, evB.v[pi - 1], 0), __func__, __LINE__)) { rc = 1; break; }
      if (launch_dgemm(false, true, trailing, pw1, (long)pw, -1.0, p21, N, p21, N, 1.0, a + (K + pw) + (K + pw) * n, N, false, st)) { rc = 1; break; }
      // B(pi): everything to the right of the next panel, lower tiles only
      const long far = trailing - pw1;
      if (far > 0) {
        const double *p31 = p21 + pw1;
        if (!check_hip(hipStreamWaitEvent(sB.s, evF.v[pi], 0), __func__, __LINE__)) { rc = 1; break; }
        if (launch_dgemm(false, true, far, far, (long)pw, -1.0, p31, N, p31, N, 1.0, a + (K + pw + pw1) + (K + pw + pw1) * n, N, true, sB.s)) { rc = 1; break; }
      }
      if (!check_hip(hipEventRecord(evB.v[pi], sB.s), __func__, __LINE__)) { rc = 1; break; }
    }
    if (!check_hip(hipStreamSynchronize(sB.s), __func__, __LINE__) || !check_hip(hipStreamSynchronize(sA.s), __func__, __LINE__)) rc = 1;
    st = nullptr;   // the solves below: legacy default stream (both streams are idle and are destroyed at the end of this scope)
    if (rc) break;
    if (verbose) { const auto t1 = tick(); debug_info("Time for potrf: %.3fs", secs(t_start, t1)); t_start = t1; }
    int info = 0;
    if (!check_hip(hipMemcpy(&info, dInfo.p, sizeof(int), hipMemcpyDeviceToHost), __func__, __LINE__)) { rc = 1; break; }
    if (info != 0) {   // wording of the reference (solve_cuda.cu:196-199); no device reset here
      set_error(22, "Error: Cholesky factorization failed at minor %d", info); rc = 1; break;
    }
    // A X = B:  L Y = B, then L^T X = Y, block by block with the inverted diagonal blocks (L is read once per solve)
    for (long kb = 0; kb < nblk && !rc; kb++) {
      const long off = kb * kPotrfNB, nb = std::min<long>(kPotrfNB, N - off), rest = N - off - nb;
      if (launch_dgemm(false, false, nb, (long)nrhs, nb, 1.0, inv + kb * kPotrfNB * kPotrfNB, kPotrfNB, b + off, N, 0.0, b + off, N, false, st)) rc = 1;
      if (!rc && rest > 0 && launch_dgemm(false, false, rest, (long)nrhs, nb, -1.0, a + (off + nb) + off * n, N, b + off, N, 1.0, b + off + nb, N, false, st)) rc = 1;
    }
    for (long kb = nblk - 1; kb >= 0 && !rc; kb--) {
      const long off = kb * kPotrfNB, nb = std::min<long>(kPotrfNB, N - off);
      if (launch_dgemm(true, false, nb, (long)nrhs, nb, 1.0, inv + kb * kPotrfNB * kPotrfNB, kPotrfNB, b + off, N, 0.0, b + off, N, false, st)) rc = 1;
      if (!rc && off > 0 && launch_dgemm(true, false, off, (long)nrhs, nb, -1.0, a + off, N, b + off, N, 1.0, b, N, false, st)) rc = 1;
    }
    if (rc) break;
    if (verbose) { const auto t1 = tick(); debug_info("Time for potrs: %.3fs", secs(t_start, t1)); t_start = t1; }
    if (logdet) {
      if (!check_hip(hipMalloc(&dLog.p, sizeof(double)), __func__, __LINE__)) { rc = 1; break; }
      hipLaunchKernelGGL(k_logdet, dim3(1), dim3(1024), 0, nullptr, static_cast<const double *>(dA.p), (long)n, static_cast<double *>(dLog.p));
      if (!check_hip(hipGetLastError(), __func__, __LINE__)) { rc = 1; break; }
      if (!check_hip(hipMemcpy(logdet, dLog.p, sizeof(double), solve_is_device_ptr(logdet) ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost), __func__, __LINE__)) { rc = 1; break; }
    }
    if (!check_hip(hipMemcpy(X, dB.p, sizeof(double) * n * nrhs, solve_is_device_ptr(X) ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost), __func__, __LINE__)) { rc = 1; break; }
  } while (0);
  (void)hipDeviceSynchronize();
  return rc;
}

// ------------------------------------------------------------------------------------------------ sparse triangular solve
constexpr uint32_t kSparseMagic = 0x4d585331u;   // "MXS1"
constexpr int kSpRhs = 8;                        // right-hand sides per pass over a row's entries

struct SparseSolve {
  uint32_t magic = kSparseMagic;
  int device = 0;
  long m = 0, nnz = 0, ncol = 0;
  int is_lower = 0;
  // [0]: A as given, [1]: A^T (the reference builds the second COO descriptor with I and J swapped, solve_cuda.cu:399-410)
  int64_t *d_rowptr[2] = {nullptr, nullptr}, *d_col[2] = {nullptr, nullptr};
  double *d_val[2] = {nullptr, nullptr};
  double *d_B = nullptr, *d_X = nullptr;
  int *d_flag = nullptr, *d_err = nullptr;
};

static void destroy_sparse(SparseSolve *s) {
  if (!s) return;
  (void)hipSetDevice(s->device);
  (void)hipDeviceSynchronize();
  for (int t = 0; t < 2; t++) {
    void *ptrs[] = {s->d_rowptr[t], s->d_col[t], s->d_val[t]};
    for (void *p : ptrs) if (p) (void)hipFree(p);
  }
  void *ptrs[] = {s->d_B, s->d_X, s->d_flag, s->d_err};
  for (void *p : ptrs) if (p) (void)hipFree(p);
  s->magic = 0;
  delete s;
}

// Synchronisation-free triangular solve T X = B (T in CSR, columns ascending inside a row, diagonal present): one wave per row.
// ascending != 0 (T lower): wave w takes row w; else (T upper) row m-1-w -- a row only depends on rows taken by waves with a
// smaller index, which the dispatcher has started no later, so the waits below always end.  The lanes share the row's off-diagonal
// entries; for each they wait until the flag of that entry's column is set, read X there (device-scope loads: the value was
// written by another CU) and accumulate for up to kSpRhs right-hand sides at a time; a cross-lane sum, one division, device-scope
// stores, a fence, the row's flag.  Every wait is bounded: a wave that gives up raises *err and still sets its flag, so the grid
// drains whatever the input.
__global__ void __launch_bounds__(256) k_sptrsm(const int64_t *__restrict__ rowptr, const int64_t *__restrict__ col, const double *__restrict__ val, long m,
                                                int ascending, const double *__restrict__ B, double *X, long ncol, int *flag, int *err) {
  const long w = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (w >= m) return;
  const int lane = threadIdx.x & 63;
  const long r = ascending ? w : m - 1 - w;
  const int64_t e0 = rowptr[r], e1 = rowptr[r + 1];
  const double diag = ascending ? val[e1 - 1] : val[e0];             // checked on the host: present and non-zero
  const int64_t o0 = ascending ? e0 : e0 + 1, o1 = ascending ? e1 - 1 : e1;
  // wait for all the columns this row needs (once; the flags only ever go from 0 to 1)
  for (int64_t e = o0 + lane; e < o1; e += 64) {
    const int64_t j = col[e];
    long spins = 0;
    while (__hip_atomic_load(flag + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {   // relaxed polls: no cache invalidate per poll
      __builtin_amdgcn_s_sleep(1);
      if (++spins > (1L << 24)) { atomicExch(err, 1); break; }
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");                  // one acquire after all the flags were seen
  __builtin_amdgcn_wave_barrier();
  for (long c0 = 0; c0 < ncol; c0 += kSpRhs) {
    double acc[kSpRhs];
#pragma unroll
    for (int c = 0; c < kSpRhs; c++) acc[c] = 0.0;
    for (int64_t e = o0 + lane; e < o1; e += 64) {
      const int64_t j = col[e];
      const double v = val[e];
#pragma unroll
      for (int c = 0; c < kSpRhs; c++)
        if (c0 + c < ncol) acc[c] = fma(v, __hip_atomic_load(X + j + (c0 + c) * m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), acc[c]);
    }
#pragma unroll
    for (int c = 0; c < kSpRhs; c++) {
#pragma unroll
      for (int sh = 32; sh > 0; sh >>= 1) acc[c] += __shfl_xor(acc[c], sh);
    }
    if (lane < kSpRhs && c0 + lane < ncol) {
      double a = 0.0;
#pragma unroll
      for (int c = 0; c < kSpRhs; c++) a = lane == c ? acc[c] : a;
      const double x = (B[r + (c0 + lane) * m] - a) / diag;
      __hip_atomic_store(X + r + (c0 + lane) * m, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  __builtin_amdgcn_wave_barrier();
  if (lane == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");                // the X stores of the whole wave (same wave: program order) are visible first
    __hip_atomic_store(flag + r, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// COO (one-based, any order) -> CSR (zero-based, rows ascending, columns ascending inside a row) of A (t = 0) or A^T (t = 1)
static void coo_to_csr(const double *V, const long *I, const long *J, long nnz, long m, int t, std::vector<int64_t> &rowptr, std::vector<int64_t> &col,
                       std::vector<double> &val) {
  const long *R = t ? J : I, *C = t ? I : J;
  std::vector<long> order((size_t)nnz);
  std::iota(order.begin(), order.end(), 0L);
  std::sort(order.begin(), order.end(), [&](long a, long b) { return R[a] != R[b] ? R[a] < R[b] : C[a] < C[b]; });
  rowptr.assign((size_t)m + 1, 0);
  col.resize((size_t)nnz); val.resize((size_t)nnz);
  for (long e = 0; e < nnz; e++) rowptr[(size_t)R[e]]++;            // R is one-based: count lands in slot row+1 of the zero-based pointer
  for (long r = 0; r < m; r++) rowptr[(size_t)r + 1] += rowptr[(size_t)r];
  for (long e = 0; e < nnz; e++) { col[(size_t)e] = C[order[(size_t)e]] - 1; val[(size_t)e] = V[order[(size_t)e]]; }
}

static int sparse_init_impl(const double *V, const long *I, const long *J, long nnz, long m, long ncol, int is_lower, void **GPU_obj) {
  if (GPU_obj) *GPU_obj = nullptr;
  if (!V || !I || !J || !GPU_obj || nnz <= 0 || m <= 0 || ncol <= 0) { set_error(1, "sparse2gpu: invalid argument"); return 1; }
  // The reference only sets CUSPARSE_SPMAT_FILL_MODE on the descriptor (solve_cuda.cu:306-308, 376-412): cuSPARSE then reads the
  // selected triangle and ignores everything else, so a caller may pass a full symmetric / general matrix.  Same here: entries
  // outside the selected triangle are dropped while the CSR is built.
  std::vector<double> Vk; std::vector<long> Ik, Jk;
  Vk.reserve((size_t)nnz); Ik.reserve((size_t)nnz); Jk.reserve((size_t)nnz);
  for (long e = 0; e < nnz; e++) {
    if (I[e] < 1 || I[e] > m || J[e] < 1 || J[e] > m) { set_error(1, "sparse2gpu: entry %ld has index (%ld, %ld) outside 1..%ld (one-based COO expected)", e, I[e], J[e], m); return 1; }
    if (is_lower ? J[e] > I[e] : J[e] < I[e]) continue;
    Vk.push_back(V[e]); Ik.push_back(I[e]); Jk.push_back(J[e]);
  }
  V = Vk.data(); I = Ik.data(); J = Jk.data(); nnz = (long)Vk.size();
  if (nnz <= 0) { set_error(1, "sparse2gpu: no entry lies in the %s triangle", is_lower ? "lower" : "upper"); return 1; }
  if (solve_select_device("sparse2gpu")) return 1;
  SparseSolve *s = new SparseSolve();
  (void)hipGetDevice(&s->device);
  s->m = m; s->nnz = nnz; s->ncol = ncol; s->is_lower = is_lower;
  auto fail = [&](const char *what) { set_error(23, "sparse2gpu: %s", what); destroy_sparse(s); return 1; };
  if (hipMalloc(reinterpret_cast<void **>(&s->d_B), sizeof(double) * (size_t)m * ncol) != hipSuccess ||
      hipMalloc(reinterpret_cast<void **>(&s->d_X), sizeof(double) * (size_t)m * ncol) != hipSuccess ||
      hipMalloc(reinterpret_cast<void **>(&s->d_flag), sizeof(int) * (size_t)m) != hipSuccess ||
      hipMalloc(reinterpret_cast<void **>(&s->d_err), sizeof(int)) != hipSuccess) return fail("hipMalloc failed");
  for (int t = 0; t < 2; t++) {
    std::vector<int64_t> rowptr, col;
    std::vector<double> val;
    coo_to_csr(V, I, J, nnz, m, t, rowptr, col, val);
    // T = A (t = 0) or A^T (t = 1) is lower triangular iff is_lower != t; its diagonal entry is the last (lower) / first (upper) of the row
    const bool lower = (is_lower != 0) != (t == 1);
    for (long r = 0; r < m; r++) {
      const int64_t e0 = rowptr[(size_t)r], e1 = rowptr[(size_t)r + 1];
      const int64_t d = lower ? e1 - 1 : e0;
      if (e1 <= e0 || col[(size_t)d] != r || val[(size_t)d] == 0.0) { set_error(1, "sparse2gpu: row %ld has no non-zero diagonal entry: the matrix is singular", r + 1); destroy_sparse(s); return 1; }
      if (e1 - e0 > 1 && (lower ? col[(size_t)(e1 - 2)] == r : col[(size_t)(e0 + 1)] == r)) { set_error(1, "sparse2gpu: duplicate diagonal entry in row %ld", r + 1); destroy_sparse(s); return 1; }
    }
    if (hipMalloc(reinterpret_cast<void **>(&s->d_rowptr[t]), sizeof(int64_t) * ((size_t)m + 1)) != hipSuccess ||
        hipMalloc(reinterpret_cast<void **>(&s->d_col[t]), sizeof(int64_t) * (size_t)nnz) != hipSuccess ||
        hipMalloc(reinterpret_cast<void **>(&s->d_val[t]), sizeof(double) * (size_t)nnz) != hipSuccess) return fail("hipMalloc failed");
    if (hipMemcpy(s->d_rowptr[t], rowptr.data(), sizeof(int64_t) * ((size_t)m + 1), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(s->d_col[t], col.data(), sizeof(int64_t) * (size_t)nnz, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(s->d_val[t], val.data(), sizeof(double) * (size_t)nnz, hipMemcpyHostToDevice) != hipSuccess) return fail("hipMemcpy failed");
  }
  *GPU_obj = s;
  return 0;
}

static int sparse_compute_impl(void *GPU_obj, char transA, const double *B, long ncol, double *X) {
  SparseSolve *s = reinterpret_cast<SparseSolve *>(GPU_obj);
  if (!s || s->magic != kSparseMagic) { set_error(2, "dcsrtrsv_solve_gpu: invalid or uninitialised sparse object"); return 1; }
  int t;
  switch (transA) {   // solve_cuda.cu:732-752 ('f' counts as transposed there too)
    case 'T': case 't': case 'f': t = 1; break;
    case 'N': case 'n': t = 0; break;
    default: set_error(1, "dcsrtrsv_solve_gpu: transA must be one of N n T t (got %c)", transA); return 1;
  }
  if (!B || !X) { set_error(1, "dcsrtrsv_solve_gpu: B and X must not be NULL"); return 1; }
  if (ncol != s->ncol) {
    set_error(1, "Sparse solve interface has been initialized with %ld columns, but %ld columns are requested by the compute function.", s->ncol, ncol);
    return 1;
  }
  MXA_HIP(hipSetDevice(s->device));
  const size_t bytes = sizeof(double) * (size_t)s->m * (size_t)ncol;
  MXA_HIP(hipMemcpy(s->d_B, B, bytes, solve_is_device_ptr(B) ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice));
  MXA_HIP(hipMemset(s->d_X, 0, bytes));
  MXA_HIP(hipMemset(s->d_flag, 0, sizeof(int) * (size_t)s->m));
  MXA_HIP(hipMemset(s->d_err, 0, sizeof(int)));
  const bool lower = (s->is_lower != 0) != (t == 1);
  const long blocks = (s->m + 3) / 4;
  if (blocks > 0x7fffffffL) { set_error(3, "dcsrtrsv_solve_gpu: matrix too large for one launch"); return 1; }
  hipLaunchKernelGGL(k_sptrsm, dim3((unsigned)blocks), dim3(256), 0, nullptr, s->d_rowptr[t], s->d_col[t], s->d_val[t], s->m, lower ? 1 : 0, s->d_B, s->d_X, ncol,
                     s->d_flag, s->d_err);
  MXA_HIP(hipGetLastError());
  MXA_HIP(hipDeviceSynchronize());
  int err = 0;
  MXA_HIP(hipMemcpy(&err, s->d_err, sizeof(int), hipMemcpyDeviceToHost));
  if (err) { set_error(24, "dcsrtrsv_solve_gpu: a row waited too long for the rows it depends on"); return 1; }
  MXA_HIP(hipMemcpy(X, s->d_X, bytes, solve_is_device_ptr(X) ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost));
  return 0;
}

}  // namespace mxa

using namespace mxa;

extern "C" {

int potrs_solve(double *A, unsigned int input_size, double *B, unsigned int rhs_cols, double *X, double *logdet, int oversubscribe) {
  clear_error();
  return dense_solve_impl(A, input_size, B, rhs_cols, X, logdet, oversubscribe);
}

void potrs_solve_gpu(double *A, unsigned int input_size, double *B, unsigned int rhs_cols, double *X, double *logdet, int oversubscribe, int *status) {
  clear_error();
  const int rc = dense_solve_impl(A, input_size, B, rhs_cols, X, logdet, oversubscribe);
  if (status) *status = rc;
}

void sparse2gpu(double *V, long *I, long *J, long nnz, long m, long ncol, int is_lower, void **GPU_obj, int *status) {
  clear_error();
  const int rc = sparse_init_impl(V, I, J, nnz, m, ncol, is_lower, GPU_obj);
  if (status) *status = rc;
}

void dcsrtrsv_solve_gpu(void *GPU_obj, char transA, double *B, long ncol, double *X, int *status) {
  clear_error();
  const int rc = sparse_compute_impl(GPU_obj, transA, B, ncol, X);
  if (status) *status = rc;
}

void free_sparse_gpu(void **GPU_obj, int *status) {
  clear_error();
  if (status) *status = 0;
  if (!GPU_obj || !*GPU_obj) return;
  SparseSolve *s = reinterpret_cast<SparseSolve *>(*GPU_obj);
  if (s->magic != kSparseMagic) { set_error(2, "free_sparse_gpu: invalid sparse object"); if (status) *status = 1; return; }
  destroy_sparse(s);
  *GPU_obj = nullptr;
}

}  // extern "C"
