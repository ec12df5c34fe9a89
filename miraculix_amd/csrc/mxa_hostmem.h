// mxa_hostmem.h -- host-side helper for results that land in FRESH pageable host memory.
//
// The reference bindings hand the library a freshly allocated result (src/bindings/Julia/crossproduct.jl:56 `M = zeros(...)`; an R / Fortran
// caller does the same): none of its pages exist yet.  The runtime's device-to-host copy into pageable memory is a CPU memcpy out of its pinned
// staging buffers, so every first touch of a 4 KiB page is a page fault INSIDE the copier thread -- at config 3 (80 GB result) 2.9 s of faults
// serialised in four copiers on top of 1.8 s of pipeline (profiles/r03_crossprod_host_abi_c3.txt).  HostPrefault populates the destination in the
// background while the operand is staged and the kernel runs: madvise(MADV_POPULATE_WRITE) (Linux >= 5.14) faults the pages in WITHOUT modifying
// them -- so there is no ordering to keep against the copiers -- in blocks handed out in ascending address order (the column slabs leave in
// that order), after a MADV_HUGEPAGE hint (where transparent huge pages are available a fault then brings in 2 MiB).  Kernels without
// MADV_POPULATE_WRITE: nothing is done (the result is the same, the first call is slower).  MXA_PREFAULT_THREADS (default 12, 0 = off).
#pragma once
#include <sys/mman.h>
#include <unistd.h>

#include <atomic>
#include <cerrno>
#include <chrono>
#include <cstdint>
#include <cstdlib>
#include <thread>
#include <vector>

namespace mxa {

class HostPrefault {
 public:
  HostPrefault() = default;
  HostPrefault(const HostPrefault &) = delete;
  HostPrefault &operator=(const HostPrefault &) = delete;
  ~HostPrefault() { join(); }

  void start(void *p, size_t bytes) {
#ifdef MADV_POPULATE_WRITE
    const char *e_want = getenv("MXA_PREFAULT_THREADS");   // read per call: A/B measurements switch it between calls
    const int want = e_want ? atoi(e_want) : 12;
    if (want <= 0 || !p || bytes < ((size_t)64 << 20)) return;
    const uintptr_t page = (uintptr_t)sysconf(_SC_PAGESIZE);
    const uintptr_t lo = ((uintptr_t)p + page - 1) / page * page, hi = ((uintptr_t)p + bytes) / page * page;
    if (hi <= lo) return;
    lo_ = lo; hi_ = hi;
#ifdef MADV_HUGEPAGE
    const char *e_huge = getenv("MXA_HOST_HUGEPAGE");
    const bool huge = !e_huge || atoi(e_huge) != 0;
    if (huge) (void)madvise(reinterpret_cast<void *>(lo), hi - lo, MADV_HUGEPAGE);   // a hint; failure is fine
#endif
    t0_ = std::chrono::steady_clock::now();
    next_.store(0);
    const int hw = (int)std::thread::hardware_concurrency();
    const int n = std::max(1, std::min(want, hw > 6 ? hw - 5 : 1));   // leave room for the copier threads and the caller
    threads_ = n;
    finished_.store(0);
    for (int t = 0; t < n; t++) th_.emplace_back([this] { work(); });
#else
    (void)p; (void)bytes;
#endif
  }
  void join() {
    for (auto &t : th_) if (t.joinable()) t.join();
    if (!th_.empty()) { seconds_ = std::chrono::duration<double>(t_done_ - t0_).count(); th_.clear(); }
  }
  // Blocks until every page below `end` has been populated (or nothing more will be: not started, finished, unsupported kernel).  A copier calls it before it
  // writes a range: a memcpy that faults pages INSIDE the region the populating threads are working on crawls (the first 1 GiB slab of config 3 took 1.0-1.2 s
  // that way, profiles/r04_crossprod_host_abi_c3.txt); behind them it runs at full speed.  Blocks are handed out in ascending order and a thread takes its
  // next block only after finishing the current one, so everything below (next_ - threads) blocks is complete.
  void wait_for(const void *end) const {
    if (!threads_ || unsupported_.load()) return;
    const uintptr_t e = std::min((uintptr_t)end, hi_);
    for (;;) {
      const size_t taken = next_.load();
      const uintptr_t done = taken > (size_t)threads_ ? lo_ + (taken - (size_t)threads_) * kBlock : lo_;
      if (done >= e || finished_.load() >= threads_ || unsupported_.load()) return;
      std::this_thread::yield();
    }
  }
  // after join(): seconds from start to the last populated block (0 when nothing ran), bytes populated, threads used, 1 if the kernel lacks the call
  double seconds() const { return seconds_; }
  size_t populated() const { return done_bytes_.load(); }
  int threads() const { return threads_; }
  bool unsupported() const { return unsupported_.load(); }

 private:
  static constexpr size_t kBlock = (size_t)32 << 20;
  void work() {
#ifdef MADV_POPULATE_WRITE
    for (;;) {
      const size_t b = next_.fetch_add(1);
      const uintptr_t a = lo_ + b * kBlock;
      if (a >= hi_ || unsupported_.load()) break;
      const size_t len = std::min<size_t>(kBlock, hi_ - a);
      if (madvise(reinterpret_cast<void *>(a), len, MADV_POPULATE_WRITE) != 0) {
        if (errno == EINVAL || errno == ENOSYS) { unsupported_.store(true); break; }   // old kernel: give up quietly
        if (errno == EAGAIN || errno == EINTR) continue;                                // skip this block; the copier's own faults cover it
      } else done_bytes_.fetch_add(len);
    }
    // the LAST thread to finish leaves the stamp: one writer, ordered before join() by the thread exit (no concurrent write of a non-atomic time_point)
    if (finished_.fetch_add(1) + 1 == (int)threads_) t_done_ = std::chrono::steady_clock::now();
#endif
  }
  uintptr_t lo_ = 0, hi_ = 0;
  std::atomic<size_t> next_{0}, done_bytes_{0};
  std::atomic<bool> unsupported_{false};
  std::atomic<int> finished_{0};
  std::vector<std::thread> th_;
  std::chrono::steady_clock::time_point t0_, t_done_;
  double seconds_ = 0.0;
  int threads_ = 0;
};

}  // namespace mxa
