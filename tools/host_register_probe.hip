// host_register_probe.hip -- can a device result reach FRESH pageable host memory faster by pinning the destination piecewise (hipHostRegister) and
// letting the copy engine write into it, than through the runtime's pinned staging + CPU memcpy (what hipMemcpy to pageable memory is)?
// Measures, on `GB` GiB of fresh anonymous memory in 1-GiB chunks:
//   A  hipHostRegister of untouched chunks (the call populates and pins)
//   B  MADV_POPULATE_WRITE with T threads, then hipHostRegister of the populated chunks (pin only)
//   C  hipMemcpyAsync device -> registered chunk;   D  hipHostUnregister
//   E  the pipeline: T populating threads ahead, R registering threads, one copy stream, unregister behind the copy -- wall time for the whole range
//   F  the plain way for comparison: populate with T threads, then ONE hipMemcpy device -> pageable per chunk (runtime staging), 4 copier threads
// build: hipcc -O2 --offload-arch=gfx950 -o tools/host_register_probe tools/host_register_probe.hip -lpthread
#include <hip/hip_runtime.h>
#include <sys/mman.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static const size_t GiB = (size_t)1 << 30;

static char *fresh(size_t bytes) {
  void *p = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
  if (p == MAP_FAILED) { printf("mmap failed\n"); exit(1); }
  return static_cast<char *>(p);
}
static void populate(char *p, size_t bytes, int threads) {
  std::atomic<size_t> next{0};
  const size_t blk = (size_t)32 << 20, nb = (bytes + blk - 1) / blk;
  std::vector<std::thread> th;
  for (int t = 0; t < threads; t++) th.emplace_back([&] {
    for (;;) { const size_t b = next.fetch_add(1); if (b >= nb) break; const size_t o = b * blk; (void)madvise(p + o, std::min(blk, bytes - o), MADV_POPULATE_WRITE); }
  });
  for (auto &t : th) t.join();
}

int main(int argc, char **argv) {
  const size_t gb = argc > 1 ? (size_t)atol(argv[1]) : 8;
  const int T = argc > 2 ? atoi(argv[2]) : 12, R = argc > 3 ? atoi(argv[3]) : 2;
  const size_t chunk = GiB, bytes = gb * GiB;
  char *d = nullptr;
  CK(hipMalloc((void **)&d, chunk));
  CK(hipMemset(d, 1, chunk));
  hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  { char *w = nullptr; CK(hipHostMalloc((void **)&w, 64 << 20, 0)); CK(hipMemcpy(w, d, 64 << 20, hipMemcpyDeviceToHost)); CK(hipHostFree(w)); }   // warm the copy path

  // A: register untouched memory
  { char *p = fresh(bytes); double t0 = now();
    for (size_t o = 0; o < bytes; o += chunk) CK(hipHostRegister(p + o, chunk, hipHostRegisterDefault));
    double t1 = now();
    for (size_t o = 0; o < bytes; o += chunk) CK(hipMemcpyAsync(p + o, d, chunk, hipMemcpyDeviceToHost, s));
    CK(hipStreamSynchronize(s)); double t2 = now();
    for (size_t o = 0; o < bytes; o += chunk) CK(hipHostUnregister(p + o));
    double t3 = now();
    printf("HOSTREG A  %zu GiB fresh: hipHostRegister (populates + pins) %.2f s = %.1f GB/s | C copy into registered %.2f s = %.1f GB/s | D unregister %.2f s = %.1f GB/s\n", gb, t1 - t0,
           bytes / (t1 - t0) * 1e-9, t2 - t1, bytes / (t2 - t1) * 1e-9, t3 - t2, bytes / (t3 - t2) * 1e-9);
    munmap(p, bytes); }
  // B: populate first, then register
  { char *p = fresh(bytes); double t0 = now(); populate(p, bytes, T); double t1 = now();
    for (size_t o = 0; o < bytes; o += chunk) CK(hipHostRegister(p + o, chunk, hipHostRegisterDefault));
    double t2 = now();
    for (size_t o = 0; o < bytes; o += chunk) CK(hipHostUnregister(p + o));
    double t3 = now();
    printf("HOSTREG B  %zu GiB fresh: MADV_POPULATE_WRITE x %d threads %.2f s = %.1f GB/s | hipHostRegister of populated pages %.2f s = %.1f GB/s | unregister %.2f s = %.1f GB/s\n", gb, T, t1 - t0,
           bytes / (t1 - t0) * 1e-9, t2 - t1, bytes / (t2 - t1) * 1e-9, t3 - t2, bytes / (t3 - t2) * 1e-9);
    munmap(p, bytes); }
  // E: pipeline
  { char *p = fresh(bytes); const size_t nc = bytes / chunk;
    std::vector<std::atomic<int>> st(nc);   // 0 nothing, 1 populated, 2 registered, 3 copied
    for (auto &x : st) x.store(0);
    double t0 = now();
    std::thread pop([&] { for (size_t c = 0; c < nc; c++) { populate(p + c * chunk, chunk, T); st[c].store(1); } });
    std::atomic<size_t> next_reg{0};
    std::vector<std::thread> regs;
    for (int r = 0; r < R; r++) regs.emplace_back([&] {
      CK(hipSetDevice(0));
      for (;;) { const size_t c = next_reg.fetch_add(1); if (c >= nc) break; while (st[c].load() < 1) std::this_thread::yield();
        CK(hipHostRegister(p + c * chunk, chunk, hipHostRegisterDefault)); st[c].store(2); }
    });
    std::vector<hipEvent_t> ev(nc);
    for (size_t c = 0; c < nc; c++) CK(hipEventCreateWithFlags(&ev[c], hipEventDisableTiming));
    std::thread unreg([&] { CK(hipSetDevice(0)); for (size_t c = 0; c < nc; c++) { while (st[c].load() < 3) std::this_thread::yield(); CK(hipEventSynchronize(ev[c])); CK(hipHostUnregister(p + c * chunk)); } });
    for (size_t c = 0; c < nc; c++) { while (st[c].load() < 2) std::this_thread::yield(); CK(hipMemcpyAsync(p + c * chunk, d, chunk, hipMemcpyDeviceToHost, s)); CK(hipEventRecord(ev[c], s)); st[c].store(3); }
    CK(hipStreamSynchronize(s)); double t1 = now();
    pop.join(); for (auto &t : regs) t.join(); unreg.join(); double t2 = now();
    bool ok = true; for (size_t c = 0; c < nc; c++) if (p[c * chunk + 12345] != 1 || p[(c + 1) * chunk - 1] != 1) ok = false;
    printf("HOSTREG E  %zu GiB fresh, pipeline (populate x %d -> register x %d -> copy engine -> unregister): data there after %.2f s = %.1f GB/s, all unpinned after %.2f s = %.1f GB/s, data %s\n", gb, T, R,
           t1 - t0, bytes / (t1 - t0) * 1e-9, t2 - t0, bytes / (t2 - t0) * 1e-9, ok ? "ok" : "WRONG");
    for (auto e : ev) (void)hipEventDestroy(e);
    munmap(p, bytes); }
  // A2 / E2: R threads register FRESH chunks in parallel (each call populates and pins its chunk), first alone, then with the copy engine following in order
  for (int R2 : {4, 8, 12}) for (size_t ck : {GiB / 4, GiB}) {
    { char *p = fresh(bytes); const size_t nc = bytes / ck; std::atomic<size_t> next{0}; std::vector<std::thread> th; double t0 = now();
      for (int r = 0; r < R2; r++) th.emplace_back([&] { CK(hipSetDevice(0)); for (;;) { const size_t c = next.fetch_add(1); if (c >= nc) break; CK(hipHostRegister(p + c * ck, ck, hipHostRegisterDefault)); } });
      for (auto &t : th) t.join(); double t1 = now();
      for (size_t c = 0; c < nc; c++) CK(hipHostUnregister(p + c * ck));
      printf("HOSTREG A2 %zu GiB fresh: hipHostRegister from %d threads, %zu-MiB chunks: %.2f s = %.1f GB/s\n", gb, R2, ck >> 20, t1 - t0, bytes / (t1 - t0) * 1e-9);
      munmap(p, bytes); }
    { char *p = fresh(bytes); const size_t nc = bytes / ck; std::vector<std::atomic<int>> st(nc); for (auto &x : st) x.store(0);
      std::atomic<size_t> next{0}; std::vector<std::thread> th; double t0 = now();
      for (int r = 0; r < R2; r++) th.emplace_back([&] { CK(hipSetDevice(0)); for (;;) { const size_t c = next.fetch_add(1); if (c >= nc) break; CK(hipHostRegister(p + c * ck, ck, hipHostRegisterDefault)); st[c].store(2); } });
      for (size_t c = 0; c < nc; c++) { while (st[c].load() < 2) std::this_thread::yield(); CK(hipMemcpyAsync(p + c * ck, d, ck, hipMemcpyDeviceToHost, s)); }
      CK(hipStreamSynchronize(s)); double t1 = now();
      for (auto &t : th) t.join();
      for (size_t c = 0; c < nc; c++) CK(hipHostUnregister(p + c * ck));
      double t2 = now();
      bool ok = true; for (size_t c = 0; c < nc; c++) if (p[c * ck + 12345] != 1 || p[(c + 1) * ck - 1] != 1) ok = false;
      printf("HOSTREG E2 %zu GiB fresh: register x %d (%zu-MiB chunks) -> copy engine in order: data there after %.2f s = %.1f GB/s, unpinned after %.2f s, data %s\n", gb, R2, ck >> 20, t1 - t0,
             bytes / (t1 - t0) * 1e-9, t2 - t0, ok ? "ok" : "WRONG");
      munmap(p, bytes); }
  }
  // F: the plain way: populate ahead, runtime copies into pageable memory from 4 threads
  { char *p = fresh(bytes); const size_t nc = bytes / chunk;
    std::vector<std::atomic<int>> st(nc); for (auto &x : st) x.store(0);
    double t0 = now();
    std::thread pop([&] { for (size_t c = 0; c < nc; c++) { populate(p + c * chunk, chunk, T); st[c].store(1); } });
    std::atomic<size_t> next{0};
    std::vector<std::thread> cp;
    for (int r = 0; r < 4; r++) cp.emplace_back([&] { CK(hipSetDevice(0)); for (;;) { const size_t c = next.fetch_add(1); if (c >= nc) break; while (st[c].load() < 1) std::this_thread::yield();
      CK(hipMemcpy(p + c * chunk, d, chunk, hipMemcpyDeviceToHost)); } });
    pop.join(); for (auto &t : cp) t.join(); double t1 = now();
    printf("HOSTREG F  %zu GiB fresh, populate x %d ahead + hipMemcpy to pageable from 4 threads: %.2f s = %.1f GB/s\n", gb, T, t1 - t0, bytes / (t1 - t0) * 1e-9);
    munmap(p, bytes); }
  // G: the same destination a second time would be warm: pageable copy into existing pages, 4 threads
  { char *p = fresh(bytes); populate(p, bytes, T); const size_t nc = bytes / chunk; double t0 = now();
    std::atomic<size_t> next{0}; std::vector<std::thread> cp;
    for (int r = 0; r < 4; r++) cp.emplace_back([&] { CK(hipSetDevice(0)); for (;;) { const size_t c = next.fetch_add(1); if (c >= nc) break; CK(hipMemcpy(p + c * chunk, d, chunk, hipMemcpyDeviceToHost)); } });
    for (auto &t : cp) t.join(); double t1 = now();
    printf("HOSTREG G  %zu GiB existing pages, hipMemcpy to pageable from 4 threads: %.2f s = %.1f GB/s\n", gb, t1 - t0, bytes / (t1 - t0) * 1e-9);
    munmap(p, bytes); }
  return 0;
}
