// xcd_exchange_probe.hip -- is a fused single-pass G v = Zc (Zc^T v) feasible on MI355X?  (DESIGN.md "single-pass G v")
//
// Every packed byte is needed twice: once for t_s = sum_i z[s,i] v_i -- which needs the WHOLE row s (all individuals) -- and, once t_s is complete, once
// for y_i += z[s,i] t_s.  tools/mall_replay_probe.hip shows that a second read is only cheaper than HBM when it hits the READING XCD's own 4 MB L2
// (served by another XCD / the Infinity Cache it runs at HBM speed), so the only single-pass shape is: an XCD (32 CUs) owns a group of R SNP rows, every
// CU keeps its 1/32 of the group's bytes in LDS, the 32 CUs exchange their partial t through the XCD's L2 and each goes on with the second use out of
// its LDS.  The LDS ring holds `lag + 1` groups, i.e. the exchange may take `lag` group periods (R = 32 rows x 25 KB = 800 KB per XCD per group =
// 0.91 us at 7 TB/s chip-wide) before the stream stalls.  This probe runs exactly that memory + synchronisation skeleton with no arithmetic:
//   per group g and workgroup (one per CU): stream `bytes_per_group` (25 KiB) by LDS-DMA into ring slot g % (lag + 1);  publish a 256-byte partial
//   (sc1 stores) + one agent-scope atomic add on the XCD's counter of group g;  then for group g - lag: poll that counter until all workgroups of the
//   XCD have arrived, read their partials (32 x 256 B, sc1 loads), touch the ring slot (ds_read).
// Reported: time against the same stream without the exchange, and the time the polling lane spent waiting.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/xcd_exchange_probe tools/xcd_exchange_probe.hip ; run: tools/xcd_exchange_probe [GiB]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#include <algorithm>
using lptr_t = __attribute__((address_space(3))) void *;
__device__ __forceinline__ void dma16(const void *sbase, uint32_t voff, uint32_t lds_addr) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %0 nt" ::"s"(sbase), "v"(voff), "s"(lds_addr) : "memory", "m0");
}
__device__ __forceinline__ int hw_xcc_id() { int x; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(x)); return x & 7; }

constexpr int kUnits = 24;                 // 1-KiB DMA units per workgroup and group (24 KiB ~ 1/32 of 32 rows x 25 KB)
constexpr int kMaxLag = 5;

// ctl: [0..7] members per XCD (registration), [8..15] registration done flags; cnt[xcd][group] arrival counters; part[xcd][group % 8][member][32 doubles]
template <bool EXCHANGE>
__global__ void __launch_bounds__(256, 1) k_probe(const char *__restrict__ src, long groups, int lag, int *__restrict__ ctl, int *__restrict__ cnt, double *__restrict__ part,
                                                  unsigned long long *__restrict__ waited, int expected_per_xcd) {
  extern __shared__ __attribute__((aligned(16))) char ring[];
  __shared__ int s_member;
  __shared__ double s_sum[32];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int xcd = hw_xcc_id();
  if (threadIdx.x == 0) s_member = __hip_atomic_fetch_add(ctl + xcd, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  const int member = s_member;
  const int P = expected_per_xcd;            // the launcher checked that every XCD received this many workgroups (one per CU)
  const uint32_t lds0 = (uint32_t)(size_t)(lptr_t)ring;
  const int slots = lag + 1;
  // the workgroup's stream: groups x kUnits KiB, consecutive
  const char *my = src + ((size_t)blockIdx.x * (size_t)groups) * (kUnits * 1024);
  unsigned long long wait_ticks = 0;
  int gave_up = 0;
  double acc = 0.0;
  for (long g = 0; g < groups + lag; g++) {
    if (g < groups) {
      const int slot = (int)(g % slots);
      const char *gb = my + (size_t)g * (kUnits * 1024);
#pragma unroll
      for (int u = 0; u < kUnits / 4; u++) dma16(gb + (wave + 4 * u) * 1024, lane * 16, lds0 + slot * (kUnits * 1024) + (wave + 4 * u) * 1024);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (EXCHANGE) {
        // publish the partial of group g: 32 doubles (256 B) by 32 lanes of wave 0, sc1; then one atomic add after the stores have left
        if (wave == 0) {
          if (lane < 32) {
            double v = (double)(g + member + lane);
            double *dst = part + (((size_t)xcd * 8 + (size_t)(g & 7)) * 64 + member) * 32 + lane;
            asm volatile("global_store_dwordx2 %0, %1, off sc1" ::"v"(dst), "v"(v) : "memory");
          }
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          if (lane == 0) __hip_atomic_fetch_add(cnt + (size_t)xcd * groups + g, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
    }
    const long h = g - lag;                  // second use of group h
    if (h >= 0) {
      if (EXCHANGE) {
        if (wave == 0) {
          const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
          if (lane == 0 && !gave_up) {
            const int *c = cnt + (size_t)xcd * groups + h;
            int seen;
            unsigned spins = 0;
            do {
              asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(seen) : "v"(c) : "memory");
              if (seen < P) __builtin_amdgcn_s_sleep(1);
            } while (seen < P && ++spins < 30000u);   // bounded (a few ms): a workgroup that once waited in vain never waits again -- the grid always drains
            if (seen < P) gave_up = 1;
          }
          wait_ticks += __builtin_amdgcn_s_memrealtime() - t0;
          // all partials of the XCD for group h: P x 32 doubles; lane l < 32 sums element l over the members in fixed order
          if (lane < 32) {
            // 32 independent agent-scope loads (global_load_dwordx2 sc1), all in flight before the first use; fixed summation order
            const double *p0 = part + (((size_t)xcd * 8 + (size_t)(h & 7)) * 64) * 32 + lane;
            double v[32];
#pragma unroll
            for (int m = 0; m < 32; m++) v[m] = __hip_atomic_load(p0 + (size_t)(m < P ? m : 0) * 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            double s = 0.0;
#pragma unroll
            for (int m = 0; m < 32; m++) s += m < P ? v[m] : 0.0;
            s_sum[lane] = s;
          }
        }
        __syncthreads();
        acc += s_sum[lane & 31];
      }
      // the second use reads the retained bytes out of LDS
      const int slot = (int)(h % slots);
      const uint4 w = *reinterpret_cast<const uint4 *>(ring + slot * (kUnits * 1024) + threadIdx.x * 16);
      acc += (double)(w.x ^ w.y ^ w.z ^ w.w);
      __syncthreads();
    }
  }
  if (acc == 1.2345) waited[1024] = 1;
  if (threadIdx.x == 0) waited[blockIdx.x] = wait_ticks | ((unsigned long long)gave_up << 63);
}

// ---- second form: the stream DECOUPLED from the exchange.  Waves 1-3 only stream (two groups of DMAs in flight per wave) into a ring of R slots and
// count the landed groups in LDS; wave 0 alone publishes, polls (for the group `lag` behind), gathers, touches the retained slot and frees it.  The
// exchange latency is then hidden as long as the ring has room: R >= lag + 3.  All waits are bounded.
template <bool EXCHANGE>
__global__ void __launch_bounds__(256, 1) k_probe2(const char *__restrict__ src, long groups, int lag, int R, int *__restrict__ ctl, int *__restrict__ cnt, double *__restrict__ part,
                                                   unsigned long long *__restrict__ waited, int expected_per_xcd) {
  extern __shared__ __attribute__((aligned(16))) char ring[];
  __shared__ int s_member;
  __shared__ volatile int s_landed[8];      // per slot: streamer waves that have landed a group there (monotonic)
  __shared__ volatile int s_consumed;       // groups wave 0 is completely done with
  __shared__ volatile int s_bail;           // a bounded wait ran out somewhere: nobody waits any more (the grid drains quickly whatever happened)
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int xcd = hw_xcc_id();
  if (threadIdx.x == 0) { s_member = __hip_atomic_fetch_add(ctl + xcd, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); s_consumed = 0; s_bail = 0; }
  if (threadIdx.x < 8) s_landed[threadIdx.x] = 0;
  __syncthreads();
  const int member = s_member, P = expected_per_xcd;
  const uint32_t lds0 = (uint32_t)(size_t)(lptr_t)ring;
  const char *my = src + ((size_t)blockIdx.x * (size_t)groups) * (kUnits * 1024);
  constexpr int kPerWave = kUnits / 3;      // DMA units per streamer wave and group
  if (wave > 0) {
    // ---- streamer: group g goes to slot g % R once wave 0 has consumed group g - R
    for (long g = 0; g < groups; g++) {
      unsigned spins = 0;
      while (g >= (long)s_consumed + R && !s_bail && ++spins < 200000u) __builtin_amdgcn_s_sleep(1);
      if (spins >= 200000u) s_bail = 1;
      const int slot = (int)(g % R);
      const char *gb = my + (size_t)g * (kUnits * 1024);
#pragma unroll
      for (int u = 0; u < kPerWave; u++) dma16(gb + ((wave - 1) * kPerWave + u) * 1024, lane * 16, lds0 + slot * (kUnits * 1024) + ((wave - 1) * kPerWave + u) * 1024);
      if (g > 0) {   // the previous group of this wave has landed once only this group's units are outstanding
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kPerWave) : "memory");
        if (lane == 0) __hip_atomic_fetch_add((int *)&s_landed[(g - 1) % R], 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0) __hip_atomic_fetch_add((int *)&s_landed[(groups - 1) % R], 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    return;
  }
  // ---- wave 0: the exchange
  unsigned long long wait_ticks = 0, land_ticks = 0;
  int gave_up = 0;
  double acc = 0.0;
  for (long g = 0; g < groups + lag; g++) {
    if (g < groups) {
      const int slot = (int)(g % R), want = 3 * (int)(g / R + 1);
      const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
      unsigned spins = 0;
      while (__hip_atomic_load((int *)&s_landed[slot], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < want && !s_bail && ++spins < 200000u) __builtin_amdgcn_s_sleep(1);
      if (spins >= 200000u) { s_bail = 1; gave_up = 1; }
      land_ticks += __builtin_amdgcn_s_memrealtime() - t0;
      if (EXCHANGE) {
        if (lane < 32) {
          double v = (double)(g + member + lane);
          double *dst = part + (((size_t)xcd * 8 + (size_t)(g & 7)) * 64 + member) * 32 + lane;
          asm volatile("global_store_dwordx2 %0, %1, off sc1" ::"v"(dst), "v"(v) : "memory");
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) __hip_atomic_fetch_add(cnt + (size_t)xcd * groups + g, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    const long h = g - lag;
    if (h >= 0) {
      if (EXCHANGE) {
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        if (!gave_up) {
          const int *c = cnt + (size_t)xcd * groups + h;
          int seen;
          unsigned spins = 0;
          do {
            seen = __hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (seen < P) __builtin_amdgcn_s_sleep(1);
          } while (seen < P && ++spins < 30000u);
          if (seen < P) gave_up = 1;
        }
        wait_ticks += __builtin_amdgcn_s_memrealtime() - t0;
        if (lane < 32) {
          const double *p0 = part + (((size_t)xcd * 8 + (size_t)(h & 7)) * 64) * 32 + lane;
          double v[32];
#pragma unroll
          for (int m = 0; m < 32; m++) v[m] = __hip_atomic_load(p0 + (size_t)(m < P ? m : 0) * 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
          for (int m = 0; m < 32; m++) acc += m < P ? v[m] : 0.0;
        }
      }
      const int slot = (int)(h % R);
      const uint4 w = *reinterpret_cast<const uint4 *>(ring + slot * (kUnits * 1024) + lane * 16);
      acc += (double)(w.x ^ w.y ^ w.z ^ w.w);
      if (lane == 0) s_consumed = (int)(h + 1);
    }
  }
  if (acc == 1.2345) waited[1024] = 1;
  if (lane == 0) { waited[blockIdx.x] = wait_ticks | ((unsigned long long)gave_up << 63); waited[512 + blockIdx.x] = land_ticks; }
}

int main(int argc, char **argv) {
  const double gib = argc > 1 ? atof(argv[1]) : 6.0;
  int dev = 0; hipDeviceProp_t prop;
  (void)hipGetDevice(&dev); (void)hipGetDeviceProperties(&prop, dev);
  const int cus = prop.multiProcessorCount;
  const long groups = (long)(gib * (1ull << 30) / ((double)cus * kUnits * 1024));
  const size_t bytes = (size_t)cus * groups * kUnits * 1024;
  char *d = nullptr; int *ctl = nullptr, *cnt = nullptr; double *part = nullptr; unsigned long long *waited = nullptr;
  if (hipMalloc((void **)&d, bytes) != hipSuccess) { printf("hipMalloc failed\n"); return 1; }
  (void)hipMalloc((void **)&ctl, 64 * sizeof(int)); (void)hipMalloc((void **)&cnt, 8 * groups * sizeof(int));
  (void)hipMalloc((void **)&part, (size_t)8 * 8 * 64 * 32 * sizeof(double)); (void)hipMalloc((void **)&waited, 2048 * sizeof(unsigned long long));
  (void)hipMemset(d, 1, bytes);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  // census: how many workgroups does each XCD get with one workgroup per CU?
  (void)hipMemset(ctl, 0, 64 * sizeof(int));
  printf("XCDPROBE %d CUs, %ld groups of %d KiB per workgroup (%.2f GB in all); a group period at 7 TB/s = %.2f us\n", cus, groups, kUnits, bytes * 1e-9, (double)cus * kUnits * 1024 / 7e12 * 1e6);
  for (int lag = 1; lag <= kMaxLag; lag++) {
    const int lds = (lag + 1) * kUnits * 1024;
    if (lds > 160 * 1024) break;
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_probe<true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_probe<false>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    float best[2] = {1e30f, 1e30f};
    double wait_us = 0, wait_max = 0;
    int per_xcd[8] = {0};
    for (int rep = 0; rep < 4; rep++) {
      for (int ex = 0; ex < 2; ex++) {
        (void)hipMemset(ctl, 0, 64 * sizeof(int)); (void)hipMemset(cnt, 0, 8 * groups * sizeof(int));
        (void)hipEventRecord(e0);
        if (ex) hipLaunchKernelGGL(k_probe<true>, dim3(cus), dim3(256), lds, 0, d, groups, lag, ctl, cnt, part, waited, cus / 8);
        else hipLaunchKernelGGL(k_probe<false>, dim3(cus), dim3(256), lds, 0, d, groups, lag, ctl, cnt, part, waited, cus / 8);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (rep > 0 && ms < best[ex]) best[ex] = ms;
        if (ex && rep == 3) {
          std::vector<unsigned long long> w(cus);
          (void)hipMemcpy(w.data(), waited, cus * sizeof(unsigned long long), hipMemcpyDeviceToHost);
          int gave = 0;
          for (int i = 0; i < cus; i++) { gave += (int)(w[i] >> 63); const double us = (double)(w[i] & ~(1ull << 63)) * 0.01; wait_us += us; wait_max = std::max(wait_max, us); }
          if (gave) printf("XCDPROBE lag %d: %d workgroups gave up waiting (uneven XCD population?)\n", lag, gave);
          (void)hipMemcpy(per_xcd, ctl, 8 * sizeof(int), hipMemcpyDeviceToHost);
        }
      }
    }
    printf("XCDPROBE lag %d (LDS ring %3d KiB): stream only %.3f ms = %.2f TB/s | with the XCD exchange %.3f ms = %.2f TB/s (x%.2f) | polling lane waited %.1f us per workgroup on average, %.1f us at most, of %.1f us; workgroups per XCD %d %d %d %d %d %d %d %d\n",
           lag, lds / 1024, best[0], bytes / best[0] * 1e-9, best[1], bytes / best[1] * 1e-9, best[1] / best[0], wait_us / cus, wait_max, best[1] * 1e3,
           per_xcd[0], per_xcd[1], per_xcd[2], per_xcd[3], per_xcd[4], per_xcd[5], per_xcd[6], per_xcd[7]);
  }
  printf("XCDPROBE decoupled form: waves 1-3 stream (two groups of DMAs in flight each), wave 0 does the exchange for the group `lag` behind; ring of R slots\n");
  for (int lag = 1; lag <= 3; lag++) {
    const int R = 6, lds = R * kUnits * 1024;
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_probe2<true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_probe2<false>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    float best[2] = {1e30f, 1e30f};
    double wait_us = 0, wait_max = 0, land_us = 0;
    int gave = 0;
    for (int rep = 0; rep < 4; rep++) {
      for (int ex = 0; ex < 2; ex++) {
        (void)hipMemset(ctl, 0, 64 * sizeof(int)); (void)hipMemset(cnt, 0, 8 * groups * sizeof(int));
        (void)hipEventRecord(e0);
        if (ex) hipLaunchKernelGGL(k_probe2<true>, dim3(cus), dim3(256), lds, 0, d, groups, lag, R, ctl, cnt, part, waited, cus / 8);
        else hipLaunchKernelGGL(k_probe2<false>, dim3(cus), dim3(256), lds, 0, d, groups, lag, R, ctl, cnt, part, waited, cus / 8);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (rep > 0 && ms < best[ex]) best[ex] = ms;
        if (ex && rep == 3) {
          std::vector<unsigned long long> w(1024);
          (void)hipMemcpy(w.data(), waited, 1024 * sizeof(unsigned long long), hipMemcpyDeviceToHost);
          for (int i = 0; i < cus; i++) { gave += (int)(w[i] >> 63); const double us = (double)(w[i] & ~(1ull << 63)) * 0.01; wait_us += us; wait_max = std::max(wait_max, us); land_us += (double)w[512 + i] * 0.01; }
        }
      }
    }
    printf("XCDPROBE decoupled lag %d (ring %d x %d KiB): stream only %.3f ms = %.2f TB/s | with the XCD exchange %.3f ms = %.2f TB/s (x%.2f) | wave 0 waited for the XCD's arrivals %.1f us per workgroup on average (%.1f at most), for its own stream %.1f us, of %.1f us; gave up: %d\n",
           lag, R, kUnits, best[0], bytes / best[0] * 1e-9, best[1], bytes / best[1] * 1e-9, best[1] / best[0], wait_us / cus, wait_max, land_us / cus, best[1] * 1e3, gave);
  }
  if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 1; }
  return 0;
}
