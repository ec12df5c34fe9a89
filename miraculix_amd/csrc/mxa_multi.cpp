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
#include "mxa_rccl.h"

#include <dlfcn.h>

#include <algorithm>
#include <cmath>
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

// the calling thread's current device is put back when a multi-device entry returns (the entries switch devices while they work;
// a caller such as PyTorch keeps its own notion of the current device)
struct DeviceRestore {
  int prev = -1;
  DeviceRestore() { if (hipGetDevice(&prev) != hipSuccess) { (void)hipGetLastError(); prev = -1; } }
  ~DeviceRestore() { if (prev >= 0) (void)hipSetDevice(prev); }
};

struct Shard {
  Handle *h = nullptr;
  long begin = 0, end = 0;                   // SNP block
  Worker *worker = nullptr;                  // borrowed from worker_pool()
  double *d_part = nullptr; size_t cap_part = 0;   // dense indiv x n partial on the shard's device
  double *d_land = nullptr; size_t cap_land = 0;   // its landing buffer on the ROOT device (remote shards, peer-to-peer mode; RCCL: shard 0 = receive buffer)
  // copy stream on the shard's device (non-blocking): pushes the partial to the root / runs the shard's ncclReduce while the shard's own
  // stream goes on with the next product (the 'T' product of the same step)
  hipStream_t cs = nullptr;
  hipEvent_t ev_part = nullptr;              // shard stream: the partial is complete
  hipEvent_t ev_pushed = nullptr;            // copy stream: the partial has landed on the root
  hipEvent_t ev_push0 = nullptr, ev_push1 = nullptr;   // timing of the push (or of ncclReduce on shard 0's copy stream)
  bool push_pending = false;
  bool pushed_recorded = false;              // ev_pushed has been recorded at least once (the next 'N' product of this shard waits for it)
  int pushes = 0; double push_ms = 0.0;
  int peer_to_root = -1, peer_from_root = -1;
};

struct Multi {
  uint32_t magic = kMagicMulti;
  long snps = 0, indiv = 0;
  int root = 0;                              // device that holds the reduced result (device of shard 0)
  std::vector<Shard> sh;
  double *d_red = nullptr; size_t cap_red = 0;     // reduced result when C is not memory of the root device
  double *d_chk = nullptr; size_t cap_chk = 0;     // RCCL cross-check: the same partials reduced peer-to-peer
  hipStream_t root_stream = nullptr;
  hipEvent_t ev_red0 = nullptr, ev_red1 = nullptr; // around the reduction kernel; ev_red1 doubles as "the previous reduction has read all partials"
  bool red_recorded = false, red_pending = false;
  // "the result of the last reduction is where the caller asked for it": recorded on the root stream at the END of multi_reduce (after the
  // copy from d_red when C is not memory of the root device).  Every shard stream waits for it before its next product of any kind, so that
  // products issued back to back with sync = 0 are ordered like calls on one stream -- an 'N' into C followed by a 'T' that reads that C as B
  hipEvent_t ev_result = nullptr;
  bool result_recorded = false;
  const char *res_lo = nullptr, *res_hi = nullptr;   // byte range of that result: only a product that touches it has to wait (the 'T' product of a
                                                      // step reads other memory and keeps running beside the reduction of the step's 'N' product)

  int reductions = 0; double reduce_ms = 0.0;
  bool use_rccl = false, rccl_checked = false;
  double rccl_diff = -1.0;
  std::vector<ncclComm_t> comm;              // per shard (created on first use of the RCCL reduction)
  bool distinct_devices = false;
  int ndevices = 0;
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

// RCCL communicators: one rank per shard, all in this process (ncclCommInitAll); needs one device per shard
int init_rccl(Multi *m) {
  if (!m->comm.empty()) return 0;
  const int G = (int)m->sh.size();
  if (!m->distinct_devices) return 2;
  if (!rccl().ok) { set_error(17, "RCCL reduction requested but librccl.so could not be loaded (%s)", dlerror() ? dlerror() : "symbols missing"); return 1; }
  std::vector<int> devs(G);
  for (int g = 0; g < G; g++) devs[g] = m->sh[g].h->device;
  m->comm.assign(G, nullptr);
  const ncclResult_t rc = rccl().CommInitAll(m->comm.data(), G, devs.data());
  if (rc != ncclSuccess) { set_error(17, "ncclCommInitAll failed: %s", rccl().GetErrorString ? rccl().GetErrorString(rc) : "?"); m->comm.clear(); return 1; }
  return 0;
}

int finish_setup(Multi *m) {
  const int G = (int)m->sh.size();
  m->root = m->sh[0].h->device;
  m->distinct_devices = true;
  std::vector<int> seen;
  for (int g = 0; g < G; g++) {
    const int d = m->sh[g].h->device;
    if (std::find(seen.begin(), seen.end(), d) != seen.end()) m->distinct_devices = false; else seen.push_back(d);
  }
  m->ndevices = (int)seen.size();
  for (int g = 0; g < G; g++) {
    Shard &S = m->sh[g];
    MXA_HIP(hipSetDevice(S.h->device));
    MXA_HIP(hipStreamCreateWithFlags(&S.cs, hipStreamNonBlocking));
    MXA_HIP(hipEventCreateWithFlags(&S.ev_part, hipEventDisableTiming));
    MXA_HIP(hipEventCreateWithFlags(&S.ev_pushed, hipEventDisableTiming));
    MXA_HIP(hipEventCreate(&S.ev_push0));
    MXA_HIP(hipEventCreate(&S.ev_push1));
  }
  // Peer access, explicit and recorded: root <-> every shard device (partials are pushed to the root, operands that live on one device are
  // read from it) and between all shard devices (operands may live on any of them).  Without it hipMemcpyPeerAsync still works, but
  // is staged through host memory instead of crossing xGMI directly.
  for (int a : seen) for (int b : seen) if (a != b) (void)enable_peer(a, b);
  std::string verdicts;
  for (int g = 0; g < G; g++) {
    Shard &S = m->sh[g];
    const int d = S.h->device;
    S.peer_to_root = d == m->root ? -1 : enable_peer(d, m->root);
    S.peer_from_root = d == m->root ? -1 : enable_peer(m->root, d);
    char buf[64];
    snprintf(buf, sizeof(buf), " %d:dev%d(%d/%d)", g, d, S.peer_to_root, S.peer_from_root);
    verdicts += buf;
  }
  MXA_HIP(hipSetDevice(m->root));
  MXA_HIP(hipStreamCreateWithFlags(&m->root_stream, hipStreamDefault));
  MXA_HIP(hipEventCreate(&m->ev_red0));
  MXA_HIP(hipEventCreate(&m->ev_red1));
  MXA_HIP(hipEventCreateWithFlags(&m->ev_result, hipEventDisableTiming));
  const char *red = getenv("MXA_REDUCE");
  if (red && std::string(red) == "rccl") {
    const int rc = init_rccl(m);
    if (rc == 1) return 1;
    if (rc == 2) debug_info("MXA_REDUCE=rccl ignored: several shards share a device (RCCL needs one rank per device); using the peer-to-peer reduction");
    else m->use_rccl = true;
  }
  debug_info("multi-device object: %d SNP shards on %d device(s), root device %d, reduction %s; shard:device(peer access to root / from root; -1 = same device):%s", G,
             m->ndevices, m->root, m->use_rccl ? "RCCL ncclReduce" : "peer-to-peer, fixed order", verdicts.c_str());
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
  // everything in flight ends first (asynchronous products, pushes, the reduction)
  for (Shard &S : m->sh) if (S.h) { (void)hipSetDevice(S.h->device); (void)hipStreamSynchronize(S.h->stream); if (S.cs) (void)hipStreamSynchronize(S.cs); }
  if (m->root_stream) { (void)hipSetDevice(m->root); (void)hipStreamSynchronize(m->root_stream); }
  for (ncclComm_t c : m->comm) if (c) (void)rccl().CommDestroy(c);
  // every shard is released by the thread that worked on it
  for (Shard &S : m->sh) {
    if (!S.h || !S.worker) continue;
    Shard *sp = &S;
    S.worker->submit([sp] {
      (void)hipSetDevice(sp->h->device);
      if (sp->d_part) (void)hipFree(sp->d_part);
      for (hipEvent_t e : {sp->ev_part, sp->ev_pushed, sp->ev_push0, sp->ev_push1}) if (e) (void)hipEventDestroy(e);
      if (sp->cs) (void)hipStreamDestroy(sp->cs);
      destroy_handle(sp->h);
      return 0;
    });
  }
  for (Shard &S : m->sh) if (S.h && S.worker) (void)S.worker->wait();
  for (Shard &S : m->sh) if (S.worker) { worker_pool().give_back(S.worker); S.worker = nullptr; }   // idle: every job has been waited for
  (void)hipSetDevice(m->root);
  for (Shard &S : m->sh) if (S.d_land) (void)hipFree(S.d_land);
  if (m->d_red) (void)hipFree(m->d_red);
  if (m->d_chk) (void)hipFree(m->d_chk);
  if (m->ev_red0) (void)hipEventDestroy(m->ev_red0);
  if (m->ev_red1) (void)hipEventDestroy(m->ev_red1);
  if (m->ev_result) (void)hipEventDestroy(m->ev_result);
  if (m->root_stream) (void)hipStreamDestroy(m->root_stream);
  m->magic = 0;
  delete m;
}

static int multi_build(long snps, long indiv, int max_n, int staging, int shards, void **out, const std::function<int(int g, long b, long e, int dev, void **h)> &make) {
  if (out) *out = nullptr;
  if (!out) { set_error(1, "plink2compressed: compressed is NULL"); return 1; }
  if (snps <= 0 || indiv <= 0) { set_error(1, "plink2compressed: snps and indiv must be positive"); return 1; }
  DeviceRestore restore;
  Multi *m = new Multi();
  m->snps = snps; m->indiv = indiv;
  std::vector<long> b, e;
  shard_blocks(snps, shards, b, e);
  const int G = (int)b.size();
  std::vector<int> dev;
  if (pick_devices(G, dev)) { delete m; return 1; }
  m->sh.resize(G);
  for (int g = 0; g < G; g++) { m->sh[g].begin = b[g]; m->sh[g].end = e[g]; m->sh[g].worker = worker_pool().borrow(); }
  // One or two packed copies (MXA_SINGLE_ORIENTATION, mxa_internal.h): decided here for the whole object, so that all shards are alike -- under the
  // automatic policy the SNP-major copy alone when, on any device, the shards it will hold do not fit with two copies (counting what their staging holds
  // while it runs) but do with one.
  int single = single_orientation_policy();
  if (single == 2) {
    single = 0;
    bool one_fits = true;
    for (int g = 0; g < G; g++) {
      bool first = true;
      for (int q = 0; q < g; q++) if (dev[q] == dev[g]) first = false;
      if (!first) continue;
      size_t two = 0, one = 0, free_b = 0, total_b = 0;
      for (int q = g; q < G; q++) if (dev[q] == dev[g]) {
        const long rows = e[q] - b[q];
        // temporaries of the staging: from a .bed file (staging 1) each shard's raw block, and its transpose for two copies; one-pointer shape (2): the raw transpose
        const size_t raw = staging == 1 ? (size_t)rows * (((size_t)indiv + 3) / 4) : 0, raw_t = staging ? (size_t)indiv * (((size_t)rows + 3) / 4) : 0;
        two += object_footprint(rows, indiv, max_n, false) + raw + raw_t;
        one += object_footprint(rows, indiv, max_n, true) + raw;
      }
      if (hipSetDevice(dev[g]) != hipSuccess || hipMemGetInfo(&free_b, &total_b) != hipSuccess) { (void)hipGetLastError(); continue; }
      if (two > free_b) single = 1;
      if (one > free_b) one_fits = false;
    }
    if (!one_fits) single = 0;   // nothing fits: let the shards' own pre-flight report it with the two-copy figure, like the reference
  }
  // stage all shards side by side: every worker uploads over its own GPU's PCIe link
  for (int g = 0; g < G; g++) {
    m->sh[g].worker->submit([&, g] {
      void *h = nullptr;
      tl_single_override = single;
      const int rc = make(g, m->sh[g].begin, m->sh[g].end, dev[g], &h);
      tl_single_override = -1;
      m->sh[g].h = reinterpret_cast<Handle *>(h);
      return rc;
    });
  }
  int rc = 0;
  for (int g = 0; g < G; g++) rc |= m->sh[g].worker->wait();
  for (int g = 0; g < G && !rc; g++) if (!m->sh[g].h) rc = 1;
  if (!rc) rc = finish_setup(m);
  if (rc) {
    if (m->sh[0].h) m->root = m->sh[0].h->device;
    multi_destroy(m);
    return 1;
  }
  *out = m;
  return 0;
}

int multi_create(const uint8_t *plink, const uint8_t *plink_t, long snps, long indiv, const double *f, int max_n, int shards, void **out) {
  if (!plink) { if (out) *out = nullptr; set_error(1, "plink2compressed: plink is NULL"); return 1; }
  const bool one_pointer = !plink_t || plink_t == plink;   // the reference's CPU call shape (benchmark.f90:185): every shard transposes its own SNP block on its device
  const size_t ps = ((size_t)indiv + 3) / 4, pi = ((size_t)snps + 3) / 4;
  return multi_build(snps, indiv, max_n, one_pointer ? 2 : 0, shards, out, [=](int, long b, long e, int dev, void **h) {
    // rows [b, e) of the SNP-major matrix; byte columns [b/4, ..) of the individual-major matrix (row pitch of the FULL matrix)
    return create_handle(plink + (size_t)b * ps, ps, one_pointer ? nullptr : plink_t + (size_t)b / 4, pi, e - b, indiv, f ? f + b : nullptr, max_n, h, dev);
  });
}

int multi_create_from_bed(const char *base, long snps, long indiv, int max_n, int shards, void **out, double *f_out) {
  const std::string b0(base);
  return multi_build(snps, indiv, max_n, 1, shards, out, [=](int, long b, long e, int dev, void **h) {
    return bed_range_to_handle(b0.c_str(), snps, indiv, b, e, max_n, dev, h, f_out ? f_out + b : nullptr);
  });
}

int multi_single(const void *obj) {
  const Multi *m = as_multi(const_cast<void *>(obj));
  return m && !m->sh.empty() && m->sh[0].h ? (m->sh[0].h->single ? 1 : 0) : -1;
}

void multi_freq(void *obj, double *f) {
  Multi *m = as_multi(obj);
  if (!m) return;
  for (const Shard &S : m->sh) memcpy(f + S.begin, S.h->h_f, sizeof(double) * (size_t)(S.end - S.begin));
}

static void harvest_push(Shard &S) {
  if (!S.push_pending) return;
  S.push_pending = false;
  float ms = 0.f;
  if (hipEventSynchronize(S.ev_push1) != hipSuccess || hipEventElapsedTime(&ms, S.ev_push0, S.ev_push1) != hipSuccess) { (void)hipGetLastError(); return; }
  S.pushes += 1; S.push_ms += ms;
}
static void harvest_reduce(Multi *m) {
  if (!m->red_pending) return;
  m->red_pending = false;
  float ms = 0.f;
  if (hipEventSynchronize(m->ev_red1) != hipSuccess || hipEventElapsedTime(&ms, m->ev_red0, m->ev_red1) != hipSuccess) { (void)hipGetLastError(); return; }
  m->reductions += 1; m->reduce_ms += ms;
}

// does the column-major operand p (rows x n, leading dimension ld) overlap the memory the last reduction delivered its result to?
static bool touches_last_result(const Multi *m, const void *p, long rows, long ld, int n) {
  if (!m->result_recorded || !p) return false;
  const char *lo = reinterpret_cast<const char *>(p), *hi = lo + sizeof(double) * ((size_t)ld * (size_t)(n - 1) + (size_t)rows);
  return lo < m->res_hi && m->res_lo < hi;
}

static int run_all(Multi *m, const std::function<int(int)> &job) {
  const int G = (int)m->sh.size();
  for (int g = 0; g < G; g++) m->sh[g].worker->submit([&job, g] { return job(g); });
  int rc = 0;
  for (int g = 0; g < G; g++) rc |= m->sh[g].worker->wait();
  return rc;
}

static int prepare_partials(Multi *m, long rows, int n) {
  for (Shard &S : m->sh) {
    if (grow_on(S.h->device, &S.d_part, &S.cap_part, (size_t)rows * n)) return 1;
    if (S.h->device != m->root && grow_on(m->root, &S.d_land, &S.cap_land, (size_t)rows * n)) return 1;   // also the scratch of the RCCL cross-check
  }
  if (m->use_rccl && grow_on(m->root, &m->sh[0].d_land, &m->sh[0].cap_land, (size_t)rows * n)) return 1;   // receive buffer of ncclReduce
  return 0;
}

// Worker of shard g, after its product was enqueued on the shard's stream: mark the partial ready and (peer-to-peer mode, remote shard)
// push it to the root over the shard's own xGMI link, on the COPY stream -- the shard's stream is free for the next product at once.
static int publish_partial(Multi *m, int g, long rows, int n, bool push) {
  Shard &S = m->sh[g];
  Handle *h = S.h;
  MXA_HIP(hipEventRecord(S.ev_part, h->stream));
  if (push && h->device != m->root) {
    harvest_push(S);
    MXA_HIP(hipStreamWaitEvent(S.cs, S.ev_part, 0));
    MXA_HIP(hipEventRecord(S.ev_push0, S.cs));
    MXA_HIP(hipMemcpyPeerAsync(S.d_land, m->root, S.d_part, h->device, sizeof(double) * (size_t)rows * n, S.cs));
    MXA_HIP(hipEventRecord(S.ev_push1, S.cs));
    MXA_HIP(hipEventRecord(S.ev_pushed, S.cs));
    S.push_pending = true; S.pushed_recorded = true;
  }
  return 0;
}

// peer-to-peer reduction of the published partials into dC (memory of the root device), ascending shard order, on the root stream
static int reduce_p2p(Multi *m, long rows, int n, double *dC, long ldc, long fill_rows, bool timed) {
  const int G = (int)m->sh.size();
  MXA_HIP(hipSetDevice(m->root));
  PartList pl{}; pl.count = G;
  for (int g = 0; g < G; g++) {
    Shard &S = m->sh[g];
    const bool remote = S.h->device != m->root;
    pl.p[g] = remote ? S.d_land : S.d_part;
    MXA_HIP(hipStreamWaitEvent(m->root_stream, remote ? S.ev_pushed : S.ev_part, 0));
  }
  if (timed) { harvest_reduce(m); MXA_HIP(hipEventRecord(m->ev_red0, m->root_stream)); }
  if (launch_reduce_parts(pl, rows, n, dC, ldc, fill_rows, m->root_stream)) return 1;
  if (timed) { MXA_HIP(hipEventRecord(m->ev_red1, m->root_stream)); m->red_recorded = true; m->red_pending = true; }
  return 0;
}

// ncclReduce of the partials onto shard 0's device: every rank's call is issued from this thread inside ONE group, each on its shard's copy
// stream behind the event of its partial.  ncclGroupEnd is reached on every path (an open group would poison RCCL for the rest of the process).
static int reduce_rccl(Multi *m, long rows, int n, double *dC, long ldc, long fill_rows) {
  const int G = (int)m->sh.size();
  Rccl &r = rccl();
  Shard &S0 = m->sh[0];
  harvest_push(S0);
  hipError_t he = hipSuccess;
  for (int g = 0; g < G && he == hipSuccess; g++) {
    he = hipSetDevice(m->sh[g].h->device);
    if (he == hipSuccess) he = hipStreamWaitEvent(m->sh[g].cs, m->sh[g].ev_part, 0);
  }
  if (he == hipSuccess) he = hipSetDevice(S0.h->device);
  if (he == hipSuccess) he = hipEventRecord(S0.ev_push0, S0.cs);
  MXA_HIP(he);
  ncclResult_t rc = r.GroupStart();
  if (rc == ncclSuccess) {
    for (int g = 0; g < G && rc == ncclSuccess && he == hipSuccess; g++) {
      he = hipSetDevice(m->sh[g].h->device);
      // the receive buffer is read on the root rank only; the other ranks pass a valid pointer all the same (their own partial)
      if (he == hipSuccess) rc = r.Reduce(m->sh[g].d_part, g == 0 ? S0.d_land : m->sh[g].d_part, (size_t)rows * n, ncclFloat64, ncclSum, 0, m->comm[g], m->sh[g].cs);
    }
    const ncclResult_t rc_end = r.GroupEnd();   // always
    if (rc == ncclSuccess) rc = rc_end;
  }
  if (rc != ncclSuccess) { set_error(17, "ncclReduce failed: %s", r.GetErrorString ? r.GetErrorString(rc) : "?"); return 1; }
  MXA_HIP(he);
  // every rank's ncclReduce reads its own partial on its own copy stream: the shard's next 'N' product waits for THAT (the root's completion
  // alone does not order the other ranks' kernels)
  for (int g = 1; g < G; g++) {
    MXA_HIP(hipSetDevice(m->sh[g].h->device));
    MXA_HIP(hipEventRecord(m->sh[g].ev_pushed, m->sh[g].cs));
    m->sh[g].pushed_recorded = true;
  }
  MXA_HIP(hipSetDevice(m->root));
  MXA_HIP(hipEventRecord(S0.ev_push1, S0.cs));
  MXA_HIP(hipEventRecord(S0.ev_pushed, S0.cs));
  S0.push_pending = true; S0.pushed_recorded = true;
  MXA_HIP(hipStreamWaitEvent(m->root_stream, S0.ev_pushed, 0));
  harvest_reduce(m);
  MXA_HIP(hipEventRecord(m->ev_red0, m->root_stream));
  PartList pl{}; pl.count = 1; pl.p[0] = S0.d_land;
  if (launch_reduce_parts(pl, rows, n, dC, ldc, fill_rows, m->root_stream)) return 1;
  MXA_HIP(hipEventRecord(m->ev_red1, m->root_stream));
  m->red_recorded = true; m->red_pending = true;
  return 0;
}

// First RCCL reduction of an object: the same partials are also pushed peer-to-peer and added in fixed order, and the two sums are
// compared on the host (<= 1e-13 of the largest entry; they differ only in the summation order).  This is the multi-rank evidence for
// the grouped ncclReduce: a mismatch fails the call.
static int rccl_cross_check(Multi *m, long rows, int n) {
  const int G = (int)m->sh.size();
  if (grow_on(m->root, &m->d_chk, &m->cap_chk, (size_t)rows * n)) return 1;
  for (int g = 0; g < G; g++) {
    Shard &S = m->sh[g];
    if (S.h->device == m->root) continue;
    MXA_HIP(hipSetDevice(S.h->device));
    MXA_HIP(hipMemcpyPeerAsync(S.d_land, m->root, S.d_part, S.h->device, sizeof(double) * (size_t)rows * n, S.cs));
    MXA_HIP(hipEventRecord(S.ev_pushed, S.cs));
    S.pushed_recorded = true;
  }
  if (reduce_p2p(m, rows, n, m->d_chk, rows, rows, false)) return 1;
  MXA_HIP(hipStreamSynchronize(m->root_stream));
  std::vector<double> a((size_t)rows * n), b((size_t)rows * n);
  MXA_HIP(hipMemcpy(a.data(), m->sh[0].d_land, sizeof(double) * a.size(), hipMemcpyDeviceToHost));
  MXA_HIP(hipMemcpy(b.data(), m->d_chk, sizeof(double) * b.size(), hipMemcpyDeviceToHost));
  double dmax = 0.0, amax = 0.0;
  bool bad = false;
  for (size_t i = 0; i < a.size(); i++) {
    if (a[i] != a[i] || b[i] != b[i]) { if ((a[i] != a[i]) != (b[i] != b[i])) bad = true; continue; }
    dmax = std::max(dmax, std::fabs(a[i] - b[i])); amax = std::max(amax, std::fabs(b[i]));
  }
  m->rccl_diff = amax > 0.0 ? dmax / amax : dmax;
  m->rccl_checked = true;
  debug_info("RCCL reduction cross-checked against the peer-to-peer fixed-order reduction: max rel diff %.3e", m->rccl_diff);
  if (bad || !(m->rccl_diff <= 1e-13)) {
    set_error(18, "RCCL reduction disagrees with the peer-to-peer fixed-order reduction (max rel diff %.3e > 1e-13): results are not trusted", m->rccl_diff);
    return 1;
  }
  return 0;
}

// reduce the shards' dense rows x n partials (d_part, complete when ev_part fires) into C (host or any device)
static int multi_reduce(Multi *m, long rows, int n, double *C, long ldc, bool sync) {
  MXA_HIP(hipSetDevice(m->root));
  int c_devno = -1;
  const bool c_dev = ptr_location(C, &c_devno) == 1, c_local = c_dev && c_devno == m->root;
  double *dC = C; long dldc = ldc;
  if (!c_local) {
    if (grow_on(m->root, &m->d_red, &m->cap_red, (size_t)ldc * n)) return 1;
    dC = m->d_red;
  }
  if (m->use_rccl) {
    if (reduce_rccl(m, rows, n, dC, dldc, ldc)) return 1;
    if (!m->rccl_checked && rccl_cross_check(m, rows, n)) return 1;
  } else if (reduce_p2p(m, rows, n, dC, dldc, ldc, true)) return 1;
  MXA_HIP(hipSetDevice(m->root));
  if (!c_local) MXA_HIP(hipMemcpyAsync(C, dC, sizeof(double) * (size_t)ldc * n, hipMemcpyDefault, m->root_stream));
  MXA_HIP(hipEventRecord(m->ev_result, m->root_stream));
  m->result_recorded = true;
  m->res_lo = reinterpret_cast<const char *>(C);
  m->res_hi = m->res_lo + sizeof(double) * ((size_t)ldc * (size_t)(n - 1) + (size_t)std::max(rows, ldc));
  if (sync || !c_dev) MXA_HIP(hipStreamSynchronize(m->root_stream));
  return 0;
}

static void harvest_all(Multi *m) {
  for (Shard &S : m->sh) { (void)hipSetDevice(S.h->device); harvest_profile(S.h); harvest_push(S); }
  (void)hipSetDevice(m->root);
  harvest_reduce(m);
}

int multi_synchronize(void *obj) {
  Multi *m = as_multi(obj);
  if (!m) { set_error(2, "mxa_multi_synchronize: not a multi-device object"); return 1; }
  DeviceRestore restore;
  for (Shard &S : m->sh) { MXA_HIP(hipSetDevice(S.h->device)); MXA_HIP(hipStreamSynchronize(S.h->stream)); MXA_HIP(hipStreamSynchronize(S.cs)); }
  MXA_HIP(hipSetDevice(m->root));
  MXA_HIP(hipStreamSynchronize(m->root_stream));
  harvest_all(m);
  return 0;
}

// One product on a multi-device object.  Operands either as ONE pointer each (the plain ABI: B / C host memory or memory of any device) or
// per shard (Bs / Cs: shard g's own slice, normally memory of shard g's device -- nothing crosses a device boundary but the partials):
//   'N': Bs[g] = rows [begin_g, end_g) of B (ld ldb);  the reduced result goes to Cs[0] (ld ldc; memory of any device or host)
//   'T': Bs[g] = the whole B (indiv x n, ld ldb) as shard g sees it (a NULL entry: Bs[0] is read across devices);  Cs[g] = rows
//        [begin_g, end_g) of C (ld ldc)
// sync == 0 (device operands only): returns when everything is enqueued -- the next product may be issued at once (the 'T' product of a step
// then runs on the shard streams while the partials of 'N' travel and are added on the root); mxa_multi_synchronize() waits.
static int multi_product(Multi *m, bool trans, int n, const double *B, const double *const *Bs, long ldb, double *C, double *const *Cs, long ldc, bool sync) {
  if (n <= 0) return 0;
  DeviceRestore restore;
  const long snps = m->snps, indiv = m->indiv;
  const int G = (int)m->sh.size();
  const bool per_shard = Bs != nullptr;
  if (per_shard) {
    if (!Cs || !Bs[0] || !Cs[0]) { set_error(1, "mxa_dgemm_compressed_multi: B_per_shard[0] and C_per_shard[0] must not be NULL"); return 1; }
    for (int g = 0; g < G; g++) if ((trans && !Cs[g]) || (!trans && !Bs[g])) { set_error(1, "mxa_dgemm_compressed_multi: shard %d has no %s slice", g, trans ? "C" : "B"); return 1; }
  } else if (!B || !C) { set_error(1, "dgemm_compressed: B and C must not be NULL"); return 1; }
  const long k_all = trans ? indiv : snps, m_all = trans ? snps : indiv;
  if (!per_shard && (ldb < k_all || ldc < m_all)) { set_error(7, "dgemm_compressed: leading dimension too small (ldb %ld < %ld or ldc %ld < %ld)", ldb, k_all, ldc, m_all); return 1; }
  int rc;
  if (trans) {
    // C (snps x n): shard g owns rows [begin_g, end_g); with ONE C the last shard also zero-fills the ld padding rows [snps, ldc)
    rc = run_all(m, [&](int g) {
      Shard &S = m->sh[g];
      const long rows = S.end - S.begin;
      const double *Bg = per_shard ? (Bs[g] ? Bs[g] : Bs[0]) : B;
      double *Cg = per_shard ? Cs[g] : C + S.begin;
      const long fill = per_shard ? rows : (g == G - 1 ? ldc - S.begin : rows);
      // an operand in host memory or on another GPU: synchronous inside the shard's own worker thread, so that a big one travels in row
      // ranges behind the product (gemm_host_pipelined)
      int bd = -1, cd = -1;
      const bool remote_op = ptr_location(Bg, &bd) == 0 || bd != S.h->device || ptr_location(Cg, &cd) == 0 || cd != S.h->device;
      // B may be the result of the 'N' product issued just before (sync = 0): the root stream delivers it, the shard streams are not
      // ordered against the root stream by themselves
      MXA_HIP(hipSetDevice(S.h->device));
      if (touches_last_result(m, Bg, indiv, ldb, n) || touches_last_result(m, Cg, fill, ldc, n)) MXA_HIP(hipStreamWaitEvent(S.h->stream, m->ev_result, 0));
      if (gemm_any(S.h, true, n, Bg, ldb, Cg, ldc, fill, sync || remote_op, true)) return 1;
      return 0;
    });
  } else {
    if (prepare_partials(m, indiv, n)) return 1;
    rc = run_all(m, [&](int g) {
      Shard &S = m->sh[g];
      MXA_HIP(hipSetDevice(S.h->device));
      // the previous reduction has read this shard's partial (and its landing buffer) before the product overwrites it
      if (m->red_recorded) MXA_HIP(hipStreamWaitEvent(S.h->stream, m->ev_red1, 0));
      if (S.pushed_recorded) MXA_HIP(hipStreamWaitEvent(S.h->stream, S.ev_pushed, 0));   // ... and this shard's own push / ncclReduce has read it
      const double *Bg = per_shard ? Bs[g] : B + S.begin;
      if (touches_last_result(m, Bg, S.end - S.begin, ldb, n)) MXA_HIP(hipStreamWaitEvent(S.h->stream, m->ev_result, 0));   // B is (part of) the previous result
      // a B in host memory or on another GPU: synchronous inside the shard's own worker thread, so that a big one arrives in K ranges behind
      // the product (gemm_host_pipelined)
      int bd = -1;
      const bool remote_op = ptr_location(Bg, &bd) == 0 || bd != S.h->device;
      if (gemm_any(S.h, false, n, Bg, ldb, S.d_part, indiv, indiv, remote_op, true)) return 1;
      return publish_partial(m, g, indiv, n, !m->use_rccl);
    });
    if (!rc) rc = multi_reduce(m, indiv, n, per_shard ? Cs[0] : C, ldc, sync);
  }
  if (sync && !rc) {
    for (Shard &S : m->sh) { MXA_HIP(hipSetDevice(S.h->device)); MXA_HIP(hipStreamSynchronize(S.h->stream)); }
    harvest_all(m);
  }
  return rc;
}

int multi_gemm(void *obj, bool trans, int n, const double *B, long ldb, double *C, long ldc) {
  Multi *m = as_multi(obj);
  if (!m) { set_error(2, "dgemm_compressed: invalid or uninitialised compressed object"); return 1; }
  return multi_product(m, trans, n, B, nullptr, ldb, C, nullptr, ldc, true);
}

int multi_gemm_slices(void *obj, bool trans, int n, const double *const *Bs, long ldb, double *const *Cs, long ldc, bool sync) {
  Multi *m = as_multi(obj);
  if (!m) { set_error(2, "mxa_dgemm_compressed_multi: not a multi-device object (create it under MIRACULIX_NUM_GPUS > 1)"); return 1; }
  if (!Bs || !Cs) { set_error(1, "mxa_dgemm_compressed_multi: the per-shard pointer arrays must not be NULL"); return 1; }
  if (!sync) {   // asynchronous only with device operands
    const int G = (int)m->sh.size();
    for (int g = 0; g < G; g++)
      if ((Bs[g] && ptr_location(Bs[g], nullptr) == 0) || (Cs[g] && (trans || g == 0) && ptr_location(Cs[g], nullptr) == 0)) sync = true;
  }
  return multi_product(m, trans, n, nullptr, Bs, ldb, nullptr, Cs, ldc, sync);
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
    Shard &S = m->sh[g];
    MXA_HIP(hipSetDevice(S.h->device));
    if (m->red_recorded) MXA_HIP(hipStreamWaitEvent(S.h->stream, m->ev_red1, 0));
    if (touches_last_result(m, V, indiv, ldv, n)) MXA_HIP(hipStreamWaitEvent(S.h->stream, m->ev_result, 0));
    if (S.pushed_recorded) MXA_HIP(hipStreamWaitEvent(S.h->stream, S.ev_pushed, 0));
    if (gram_any(S.h, n, V, ldv, S.d_part, indiv, false)) return 1;
    return publish_partial(m, g, indiv, n, !m->use_rccl);
  });
  if (!rc) rc = multi_reduce(m, indiv, n, out, ldo, true);
  for (Shard &S : m->sh) { (void)hipSetDevice(S.h->device); (void)hipStreamSynchronize(S.h->stream); }
  return rc;
}

int multi_set_reduction(void *obj, int kind) {
  Multi *m = as_multi(obj);
  if (!m) { set_error(2, "mxa_multi_set_reduction: not a multi-device object"); return 1; }
  if (kind != 0 && kind != 1) { set_error(1, "mxa_multi_set_reduction: kind must be 0 (peer-to-peer, fixed order) or 1 (RCCL)"); return 1; }
  DeviceRestore restore;
  if (multi_synchronize(obj)) return 1;
  if (kind == 1) {
    const int rc = init_rccl(m);
    if (rc) return rc;   // 2: not applicable (several shards share a device)
  }
  m->use_rccl = kind == 1;
  return 0;
}

}  // namespace mxa

using namespace mxa;

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
  return (int)reinterpret_cast<mxa::Multi *>(compressed)->sh.size();
}

extern "C" int mxa_dgemm_compressed_multi(char trans, void *compressed, int n, const double *const *B_per_shard, long ldb, double *const *C_per_shard, long ldc, int sync) {
  clear_error();
  bool t;
  if (trans == 'T' || trans == 't' || trans == 'Y' || trans == 'y') t = true;
  else if (trans == 'N' || trans == 'n') t = false;
  else exit(99);   // as dgemm_compressed (5codesAPI.c:73-77)
  return multi_gemm_slices(compressed, t, n, B_per_shard, ldb, C_per_shard, ldc, sync != 0);
}

extern "C" int mxa_multi_synchronize(void *compressed) { clear_error(); return multi_synchronize(compressed); }
extern "C" int mxa_multi_set_reduction(void *compressed, int kind) { clear_error(); return multi_set_reduction(compressed, kind); }

extern "C" int mxa_multi_get_info(void *compressed, mxa_multi_info *out) {
  Multi *m = as_multi(compressed);
  if (!m || !out) return 1;
  out->shards = (int)m->sh.size(); out->devices = m->ndevices; out->root_device = m->root;
  out->reduction = m->use_rccl ? 1 : 0;
  out->reductions = m->reductions; out->reduce_ms = m->reduce_ms;
  out->rccl_checked = m->rccl_checked ? 1 : 0; out->rccl_vs_p2p_max_rel_diff = m->rccl_diff;
  return 0;
}

extern "C" int mxa_multi_shard_info(void *compressed, int shard, mxa_shard_info *out) {
  Multi *m = as_multi(compressed);
  if (!m || !out || shard < 0 || shard >= (int)m->sh.size()) return 1;
  const Shard &S = m->sh[shard];
  out->device = S.h->device; out->snp_begin = S.begin; out->snp_end = S.end;
  out->peer_to_root = S.peer_to_root; out->peer_from_root = S.peer_from_root;
  out->kernel_launches = S.h->prof.launches; out->kernel_ms = S.h->prof.kernel_ms;
  out->in_copies = S.h->prof.in_copies; out->in_ms = S.h->prof.in_ms;
  out->out_copies = S.h->prof.out_copies; out->out_ms = S.h->prof.out_ms;
  out->pushes = S.pushes; out->push_ms = S.push_ms;
  return 0;
}

extern "C" int mxa_multi_reset_profile(void *compressed) {
  Multi *m = as_multi(compressed);
  if (!m) return 1;
  if (multi_synchronize(compressed)) return 1;
  for (Shard &S : m->sh) { S.h->prof = ObjectProfile(); S.pushes = 0; S.push_ms = 0.0; }
  m->reductions = 0; m->reduce_ms = 0.0;
  return 0;
}
