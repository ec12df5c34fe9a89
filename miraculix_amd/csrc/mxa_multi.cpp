// mxa_multi.cpp -- SNP-sharded objects over several devices BEHIND the reference C ABI (north_star: "host C++ owns the
// staging ... the SNP dimension shards across the GPUs of one node with an all-reduce on the (indiv x ncol) output").
//
// MIRACULIX_NUM_GPUS=G (default 1) makes plink2compressed / mxa_bed2compressed return ONE handle that owns G per-device objects over
// contiguous SNP blocks (boundaries at multiples of 4, so the packed bytes of the individual-major matrix split cleanly).  The
// Julia / Fortran / R bindings keep calling the same four symbols (reference: src/miraculix/5codesAPI.c:80-110; its device
// selection is env-driven too, src/cuda/cuda_utils.cu:187-247):
//   'N'  C(indiv x n) = sum_g Zc_g B[s_g, :]   -- every shard multiplies its SNP block with its rows of B on its own device and
//        stream (one worker thread per shard: uploads of B from pageable host memory run side by side on the GPUs' own PCIe
//        links); the indiv x n partials (centring term included: it is a partial sum too) are reduced onto the first device.
//   'T'  C(snps x n): every shard writes its own row block -- no exchange.
// Reduction.  Default: peer-to-peer -- every shard pushes its partial to the root device over its own xGMI link
// (hipMemcpyPeerAsync; the links are point-to-point, 7 pushes run in parallel) and one kernel adds them in ASCENDING SHARD ORDER, so
// the result is bitwise reproducible run to run and independent of timing.  MXA_REDUCE=rccl uses ncclReduce (RCCL, dlopen()ed,
// ncclCommInitAll in this process) instead; it needs distinct devices.  Shards beyond the visible devices are placed round-robin
// ("virtual shards": several SNP blocks on one GPU) -- that is how the path is tested on a one-GPU box.
#include "../../include/miraculix_amd.h"
#include "mxa_internal.h"

#include <dlfcn.h>

#include <algorithm>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace mxa {

namespace {

// ---- one persistent worker thread per shard: it owns the shard's device binding and runs the shard's jobs in order
class Worker {
 public:
  Worker() : th_([this] { loop(); }) {}
  ~Worker() {
    { std::lock_guard<std::mutex> lk(m_); stop_ = true; }
    cv_.notify_all();
    th_.join();
  }
  void submit(std::function<int()> job) {
    { std::lock_guard<std::mutex> lk(m_); job_ = std::move(job); has_job_ = true; done_ = false; }
    cv_.notify_all();
  }
  int wait() {
    std::unique_lock<std::mutex> lk(m_);
    cv_.wait(lk, [this] { return done_; });
    return rc_;
  }

 private:
  void loop() {
    mark_thread_concurrent();   // strided downloads issued from here take the per-column path (see copy_columns in mxa_api.cpp)
    for (;;) {
      std::function<int()> job;
      {
        std::unique_lock<std::mutex> lk(m_);
        cv_.wait(lk, [this] { return has_job_ || stop_; });
        if (stop_ && !has_job_) return;
        job = std::move(job_); has_job_ = false;
      }
      const int rc = job();
      { std::lock_guard<std::mutex> lk(m_); rc_ = rc; done_ = true; }
      cv_.notify_all();
    }
  }
  std::mutex m_;
  std::condition_variable cv_;
  std::function<int()> job_;
  bool has_job_ = false, stop_ = false, done_ = true;
  int rc_ = 0;
  std::thread th_;
};

// Worker threads are kept for the life of the process and lent to multi-device objects (a thread per shard): objects that are created and
// destroyed repeatedly reuse the same threads, and neither creation nor destruction pays for thread start-up / joins.
class WorkerPool {
 public:
  Worker *borrow() {
    std::lock_guard<std::mutex> lk(m_);
    if (!free_.empty()) { Worker *w = free_.back(); free_.pop_back(); return w; }
    all_.emplace_back(new Worker());
    return all_.back().get();
  }
  void give_back(Worker *w) { std::lock_guard<std::mutex> lk(m_); free_.push_back(w); }
 private:
  std::mutex m_;
  std::vector<std::unique_ptr<Worker>> all_;
  std::vector<Worker *> free_;
};
WorkerPool &worker_pool() { static WorkerPool *p = new WorkerPool(); return *p; }   // never destroyed: no thread joins at process exit

// ---- RCCL, bound at run time (the library does not link it; the default reduction does not need it)
struct Rccl {
  void *lib = nullptr;
  int (*CommInitAll)(void **, int, const int *) = nullptr;
  int (*CommDestroy)(void *) = nullptr;
  int (*Reduce)(const void *, void *, size_t, int, int, int, void *, hipStream_t) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  const char *(*GetErrorString)(int) = nullptr;
  bool ok = false;
};
constexpr int kNcclFloat64 = 8, kNcclSum = 0;   // rccl.h: ncclDataType_t::ncclFloat64, ncclRedOp_t::ncclSum

Rccl &rccl() {
  static Rccl r = [] {
    Rccl q;
    for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      q.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
      if (q.lib) break;
    }
    if (!q.lib) return q;
    q.CommInitAll = reinterpret_cast<decltype(q.CommInitAll)>(dlsym(q.lib, "ncclCommInitAll"));
    q.CommDestroy = reinterpret_cast<decltype(q.CommDestroy)>(dlsym(q.lib, "ncclCommDestroy"));
    q.Reduce = reinterpret_cast<decltype(q.Reduce)>(dlsym(q.lib, "ncclReduce"));
    q.GroupStart = reinterpret_cast<decltype(q.GroupStart)>(dlsym(q.lib, "ncclGroupStart"));
    q.GroupEnd = reinterpret_cast<decltype(q.GroupEnd)>(dlsym(q.lib, "ncclGroupEnd"));
    q.GetErrorString = reinterpret_cast<decltype(q.GetErrorString)>(dlsym(q.lib, "ncclGetErrorString"));
    q.ok = q.CommInitAll && q.CommDestroy && q.Reduce && q.GroupStart && q.GroupEnd;
    return q;
  }();
  return r;
}

// the calling thread's current device is put back when a multi-device entry returns (the entries switch devices while they work;
// a caller such as PyTorch keeps its own notion of the current device)
struct DeviceRestore {
  int prev = -1;
  DeviceRestore() { if (hipGetDevice(&prev) != hipSuccess) { (void)hipGetLastError(); prev = -1; } }
  ~DeviceRestore() { if (prev >= 0) (void)hipSetDevice(prev); }
};

struct Multi {
  uint32_t magic = kMagicMulti;
  long snps = 0, indiv = 0;
  int root = 0;                              // device that holds the reduced result
  std::vector<Handle *> shard;
  std::vector<long> begin, end;              // SNP block of every shard
  std::vector<Worker *> worker;              // borrowed from worker_pool()
  // per shard: dense indiv x n partial on the shard's device; on the root: one landing buffer per remote shard + the reduced result
  std::vector<double *> d_part, d_land;
  std::vector<size_t> cap_part, cap_land;
  std::vector<hipEvent_t> ev_done;           // recorded on the shard's stream when its partial (or its copy on the root) is ready
  double *d_red = nullptr; size_t cap_red = 0;
  hipStream_t root_stream = nullptr;
  bool use_rccl = false;
  std::vector<void *> comm;                  // ncclComm_t per shard (use_rccl)
};

Multi *as_multi(void *p) {
  Multi *m = reinterpret_cast<Multi *>(p);
  return (m && m->magic == kMagicMulti) ? m : nullptr;
}

int grow_on(int dev, double **p, size_t *cap, size_t elems) {
  if (*cap >= elems) return 0;
  MXA_HIP(hipSetDevice(dev));
  if (*p) { MXA_HIP(hipFree(*p)); *p = nullptr; *cap = 0; }
  MXA_HIP(hipMalloc(reinterpret_cast<void **>(p), elems * sizeof(double)));
  *cap = elems;
  return 0;
}

void shard_blocks(long snps, int want, std::vector<long> &b, std::vector<long> &e) {
  // contiguous blocks at multiples of 4, the same rule as miraculix_amd/distributed.py:shard_bounds; shards that
  // would be empty (4 * shards > snps) are dropped
  const long per = ((snps + want - 1) / want + 3) / 4 * 4;
  for (int g = 0; g < want; g++) {
    const long b0 = std::min(snps, g * per), e0 = std::min(snps, b0 + per);
    if (e0 > b0) { b.push_back(b0); e.push_back(e0); }
  }
}

int pick_devices(int nshards, std::vector<int> &dev) {
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) {
    (void)hipGetLastError();
    set_error(10, "no HIP device available. This engine is GPU-only.");
    return 1;
  }
  int base = 0;
  const char *d = getenv("HIP_DEVICE");
  if (!d) d = getenv("CUDA_DEVICE");
  if (d) base = atoi(d);
  if (base < 0 || base >= count) { set_error(11, "The requested device %d is not visible to the HIP runtime (%d devices).", base, count); return 1; }
  dev.resize(nshards);
  for (int g = 0; g < nshards; g++) dev[g] = (base + g) % count;   // more shards than devices: several SNP blocks per device
  return 0;
}

int finish_setup(Multi *m) {
  const int G = (int)m->shard.size();
  m->root = m->shard[0]->device;
  m->d_part.assign(G, nullptr); m->cap_part.assign(G, 0);
  m->d_land.assign(G, nullptr); m->cap_land.assign(G, 0);
  m->ev_done.assign(G, nullptr);
  for (int g = 0; g < G; g++) {
    MXA_HIP(hipSetDevice(m->shard[g]->device));
    MXA_HIP(hipEventCreateWithFlags(&m->ev_done[g], hipEventDisableTiming));
  }
  MXA_HIP(hipSetDevice(m->root));
  MXA_HIP(hipStreamCreateWithFlags(&m->root_stream, hipStreamDefault));
  const char *red = getenv("MXA_REDUCE");
  if (red && std::string(red) == "rccl") {
    bool distinct = true;
    for (int g = 0; g < G; g++) for (int q = 0; q < g; q++) if (m->shard[g]->device == m->shard[q]->device) distinct = false;
    if (!distinct) debug_info("MXA_REDUCE=rccl ignored: several shards share a device (RCCL needs one rank per device); using the peer-to-peer reduction");
    else if (!rccl().ok) { set_error(17, "MXA_REDUCE=rccl: librccl.so could not be loaded (%s)", dlerror() ? dlerror() : "symbols missing"); return 1; }
    else {
      std::vector<int> devs(G);
      for (int g = 0; g < G; g++) devs[g] = m->shard[g]->device;
      m->comm.assign(G, nullptr);
      const int rc = rccl().CommInitAll(m->comm.data(), G, devs.data());
      if (rc != 0) { set_error(17, "ncclCommInitAll failed: %s", rccl().GetErrorString ? rccl().GetErrorString(rc) : "?"); m->comm.clear(); return 1; }
      m->use_rccl = true;
    }
  }
  return 0;
}

}  // namespace

bool is_multi(const void *obj) { return obj && reinterpret_cast<const Multi *>(obj)->magic == kMagicMulti; }

int run_on_devices(int parts, const std::function<int(int, int)> &job) {
  if (parts <= 0) return 0;
  std::vector<int> dev;
  if (pick_devices(parts, dev)) return 1;
  DeviceRestore restore;
  std::vector<Worker *> ws;
  for (int g = 0; g < parts; g++) ws.push_back(worker_pool().borrow());
  for (int g = 0; g < parts; g++) ws[g]->submit([&job, &dev, g] { return job(g, dev[g]); });
  int rc = 0;
  for (int g = 0; g < parts; g++) { rc |= ws[g]->wait(); worker_pool().give_back(ws[g]); }
  return rc;
}

int multi_requested() {
  const char *e = getenv("MIRACULIX_NUM_GPUS");
  const int g = e ? atoi(e) : 1;
  return std::max(1, std::min(g, kMaxShards));
}

void multi_destroy(void *obj) {
  Multi *m = as_multi(obj);
  if (!m) return;
  DeviceRestore restore;
  if (m->use_rccl) for (void *c : m->comm) if (c) (void)rccl().CommDestroy(c);
  // every shard is released by the thread that worked on it
  for (size_t g = 0; g < m->shard.size(); g++) {
    if (!m->shard[g] || g >= m->worker.size()) continue;
    m->worker[g]->submit([m, g] {
      (void)hipSetDevice(m->shard[g]->device);
      if (g < m->d_part.size() && m->d_part[g]) (void)hipFree(m->d_part[g]);
      if (g < m->ev_done.size() && m->ev_done[g]) (void)hipEventDestroy(m->ev_done[g]);
      destroy_handle(m->shard[g]);
      return 0;
    });
  }
  for (size_t g = 0; g < m->shard.size(); g++) if (m->shard[g] && g < m->worker.size()) (void)m->worker[g]->wait();
  for (Worker *w : m->worker) worker_pool().give_back(w);   // idle: every job has been waited for
  m->worker.clear();
  (void)hipSetDevice(m->root);
  for (double *p : m->d_land) if (p) (void)hipFree(p);
  if (m->d_red) (void)hipFree(m->d_red);
  if (m->root_stream) (void)hipStreamDestroy(m->root_stream);
  m->magic = 0;
  delete m;
}

static int multi_build(long snps, long indiv, int shards, void **out, const std::function<int(int g, long b, long e, int dev, void **h)> &make) {
  if (out) *out = nullptr;
  if (!out) { set_error(1, "plink2compressed: compressed is NULL"); return 1; }
  if (snps <= 0 || indiv <= 0) { set_error(1, "plink2compressed: snps and indiv must be positive"); return 1; }
  DeviceRestore restore;
  Multi *m = new Multi();
  m->snps = snps; m->indiv = indiv;
  shard_blocks(snps, shards, m->begin, m->end);
  const int G = (int)m->begin.size();
  std::vector<int> dev;
  if (pick_devices(G, dev)) { delete m; return 1; }
  m->shard.assign(G, nullptr);
  for (int g = 0; g < G; g++) m->worker.push_back(worker_pool().borrow());
  // stage all shards side by side: every worker uploads over its own GPU's PCIe link
  for (int g = 0; g < G; g++) {
    m->worker[g]->submit([&, g] {
      void *h = nullptr;
      const int rc = make(g, m->begin[g], m->end[g], dev[g], &h);
      m->shard[g] = reinterpret_cast<Handle *>(h);
      return rc;
    });
  }
  int rc = 0;
  for (int g = 0; g < G; g++) rc |= m->worker[g]->wait();
  for (int g = 0; g < G && !rc; g++) if (!m->shard[g]) rc = 1;
  if (!rc) rc = finish_setup(m);
  if (rc) {
    // destroy what exists (multi_destroy needs shard[0] for the root device only when present)
    if (m->shard[0]) m->root = m->shard[0]->device;
    multi_destroy(m);
    return 1;
  }
  debug_info("multi-device object: %d SNP shards over devices starting at %d, reduction %s", G, m->root, m->use_rccl ? "RCCL ncclReduce" : "peer-to-peer, fixed order");
  *out = m;
  return 0;
}

int multi_create(const uint8_t *plink, const uint8_t *plink_t, long snps, long indiv, const double *f, int max_n, int shards, void **out) {
  if (!plink || !plink_t) { if (out) *out = nullptr; set_error(1, "plink2compressed: both plink and plink_transposed are required on the GPU path"); return 1; }
  const size_t ps = ((size_t)indiv + 3) / 4, pi = ((size_t)snps + 3) / 4;
  return multi_build(snps, indiv, shards, out, [=](int, long b, long e, int dev, void **h) {
    // rows [b, e) of the SNP-major matrix; byte columns [b/4, ..) of the individual-major matrix (row pitch of the FULL matrix)
    return create_handle(plink + (size_t)b * ps, ps, plink_t + (size_t)b / 4, pi, e - b, indiv, f ? f + b : nullptr, max_n, h, dev);
  });
}

int multi_create_from_bed(const char *base, long snps, long indiv, int max_n, int shards, void **out, double *f_out) {
  const std::string b0(base);
  return multi_build(snps, indiv, shards, out, [=](int, long b, long e, int dev, void **h) {
    return bed_range_to_handle(b0.c_str(), snps, indiv, b, e, max_n, dev, h, f_out ? f_out + b : nullptr);
  });
}

void multi_freq(void *obj, double *f) {
  Multi *m = as_multi(obj);
  if (!m) return;
  for (size_t g = 0; g < m->shard.size(); g++) memcpy(f + m->begin[g], m->shard[g]->h_f, sizeof(double) * (size_t)(m->end[g] - m->begin[g]));
}

// reduce the shards' dense m x n partials (d_part[g], complete when ev_done[g] fires) into C (host or any device), ascending order
static int multi_reduce(Multi *m, long rows, int n, double *C, long ldc) {
  const int G = (int)m->shard.size();
  MXA_HIP(hipSetDevice(m->root));
  int c_devno = -1;
  const bool c_local = ptr_location(C, &c_devno) == 1 && c_devno == m->root;
  double *dC = C; long dldc = ldc;
  if (!c_local) {
    if (grow_on(m->root, &m->d_red, &m->cap_red, (size_t)ldc * n)) return 1;
    dC = m->d_red;
  }
  if (m->use_rccl) {
    // ncclReduce into the root's landing buffer 0; every rank's call is issued from this thread inside one group
    if (grow_on(m->root, &m->d_land[0], &m->cap_land[0], (size_t)rows * n)) return 1;
    Rccl &r = rccl();
    int rc = r.GroupStart();
    for (int g = 0; g < G && !rc; g++) {
      MXA_HIP(hipSetDevice(m->shard[g]->device));
      rc = r.Reduce(m->d_part[g], g == 0 ? m->d_land[0] : nullptr, (size_t)rows * n, kNcclFloat64, kNcclSum, 0, m->comm[g], m->shard[g]->stream);
    }
    if (!rc) rc = r.GroupEnd();
    if (rc) { set_error(17, "ncclReduce failed: %s", r.GetErrorString ? r.GetErrorString(rc) : "?"); return 1; }
    MXA_HIP(hipSetDevice(m->root));
    MXA_HIP(hipEventRecord(m->ev_done[0], m->shard[0]->stream));
    MXA_HIP(hipStreamWaitEvent(m->root_stream, m->ev_done[0], 0));
    PartList pl{}; pl.count = 1; pl.p[0] = m->d_land[0];
    if (launch_reduce_parts(pl, rows, n, dC, dldc, ldc, m->root_stream)) return 1;
  } else {
    PartList pl{}; pl.count = G;
    for (int g = 0; g < G; g++) {
      pl.p[g] = m->shard[g]->device == m->root ? m->d_part[g] : m->d_land[g];
      MXA_HIP(hipStreamWaitEvent(m->root_stream, m->ev_done[g], 0));
    }
    if (launch_reduce_parts(pl, rows, n, dC, dldc, ldc, m->root_stream)) return 1;
  }
  if (!c_local) MXA_HIP(hipMemcpyAsync(C, dC, sizeof(double) * (size_t)ldc * n, hipMemcpyDefault, m->root_stream));
  MXA_HIP(hipStreamSynchronize(m->root_stream));
  return 0;
}

// after the shard's product: push the partial to the root (peer-to-peer mode, remote shard) and mark it ready
static int publish_partial(Multi *m, int g, long rows, int n) {
  Handle *h = m->shard[g];
  if (!m->use_rccl && h->device != m->root)
    MXA_HIP(hipMemcpyPeerAsync(m->d_land[g], m->root, m->d_part[g], h->device, sizeof(double) * (size_t)rows * n, h->stream));
  MXA_HIP(hipEventRecord(m->ev_done[g], h->stream));
  return 0;
}

static int prepare_partials(Multi *m, long rows, int n) {
  const int G = (int)m->shard.size();
  for (int g = 0; g < G; g++) {
    if (grow_on(m->shard[g]->device, &m->d_part[g], &m->cap_part[g], (size_t)rows * n)) return 1;
    if (!m->use_rccl && m->shard[g]->device != m->root && grow_on(m->root, &m->d_land[g], &m->cap_land[g], (size_t)rows * n)) return 1;
  }
  return 0;
}

static int run_all(Multi *m, const std::function<int(int)> &job) {
  const int G = (int)m->shard.size();
  for (int g = 0; g < G; g++) m->worker[g]->submit([&job, g] { return job(g); });
  int rc = 0;
  for (int g = 0; g < G; g++) rc |= m->worker[g]->wait();
  return rc;
}

int multi_gemm(void *obj, bool trans, int n, const double *B, long ldb, double *C, long ldc) {
  Multi *m = as_multi(obj);
  if (!m) { set_error(2, "dgemm_compressed: invalid or uninitialised compressed object"); return 1; }
  if (n <= 0) return 0;
  DeviceRestore restore;
  if (!B || !C) { set_error(1, "dgemm_compressed: B and C must not be NULL"); return 1; }
  const long snps = m->snps, indiv = m->indiv;
  const int G = (int)m->shard.size();
  int rc;
  if (trans) {
    // C (snps x n): shard g owns rows [begin_g, end_g); the last shard also zero-fills the ld padding rows [snps, ldc)
    if (ldb < indiv || ldc < snps) { set_error(7, "dgemm_compressed: leading dimension too small (ldb %ld < %ld or ldc %ld < %ld)", ldb, indiv, ldc, snps); return 1; }
    rc = run_all(m, [&](int g) {
      const long rows = m->end[g] - m->begin[g];
      const long fill = g == G - 1 ? ldc - m->begin[g] : rows;
      return gemm_any(m->shard[g], true, n, B, ldb, C + m->begin[g], ldc, fill, true, true);
    });
  } else {
    if (ldb < snps || ldc < indiv) { set_error(7, "dgemm_compressed: leading dimension too small (ldb %ld < %ld or ldc %ld < %ld)", ldb, snps, ldc, indiv); return 1; }
    if (prepare_partials(m, indiv, n)) return 1;
    rc = run_all(m, [&](int g) {
      // synchronous inside the shard's own worker thread: a big host B is then uploaded in K ranges behind the product (gemm_host_pipelined)
      if (gemm_any(m->shard[g], false, n, B + m->begin[g], ldb, m->d_part[g], indiv, indiv, true, true)) return 1;
      return publish_partial(m, g, indiv, n);
    });
    if (!rc) rc = multi_reduce(m, indiv, n, C, ldc);
    for (int g = 0; g < G; g++) { (void)hipSetDevice(m->shard[g]->device); (void)hipStreamSynchronize(m->shard[g]->stream); harvest_profile(m->shard[g]); }
  }
  return rc;
}

// out (indiv x n) = sum_g Zc_g (Zc_g^T V): every shard keeps its snps_g x n intermediate on its own device; one reduction
int multi_gram(void *obj, int n, const double *V, long ldv, double *out, long ldo) {
  Multi *m = as_multi(obj);
  if (!m) { set_error(2, "mxa_gram_matvec: invalid or uninitialised compressed object"); return 1; }
  if (n <= 0) return 0;
  DeviceRestore restore;
  if (!V || !out) { set_error(1, "mxa_gram_matvec: V and out must not be NULL"); return 1; }
  const long indiv = m->indiv;
  if (ldv < indiv || ldo < indiv) { set_error(7, "mxa_gram_matvec: leading dimension too small (ldv %ld, ldo %ld < %ld)", ldv, ldo, indiv); return 1; }
  if (prepare_partials(m, indiv, n)) return 1;
  int rc = run_all(m, [&](int g) {
    if (gram_any(m->shard[g], n, V, ldv, m->d_part[g], indiv, false)) return 1;
    return publish_partial(m, g, indiv, n);
  });
  if (!rc) rc = multi_reduce(m, indiv, n, out, ldo);
  for (size_t g = 0; g < m->shard.size(); g++) { (void)hipSetDevice(m->shard[g]->device); (void)hipStreamSynchronize(m->shard[g]->stream); }
  return rc;
}

}  // namespace mxa

extern "C" int mxa_shard_bounds(long snps, int shards, int g, long *begin, long *end) {
  std::vector<long> b, e;
  if (snps <= 0 || shards <= 0) return 0;
  mxa::shard_blocks(snps, std::min(shards, mxa::kMaxShards), b, e);
  if (g >= 0 && g < (int)b.size()) { if (begin) *begin = b[g]; if (end) *end = e[g]; }
  return (int)b.size();
}

extern "C" int mxa_num_shards(void *compressed) {
  if (!compressed) return 0;
  if (!mxa::is_multi(compressed)) return 1;
  return (int)reinterpret_cast<mxa::Multi *>(compressed)->shard.size();
}
