// tests/host/plan_sweep.cpp -- the library's host-only planners (miraculix_amd/csrc/mxa_plan.h: exactly the code the product compiles) swept over
// shapes under AddressSanitizer / UndefinedBehaviorSanitizer on the CPU.  Reference practice: an -fsanitize=address build profile
// (src/miraculix/makefile.c.mk:47-50) and valgrind targets for its test drivers (tests/dgemm_compressed/Makefile:80-85).  Never built for the GPU.
// Invariants checked: every K split is non-empty and the splits cover the K slabs exactly once, in order; the padded extents hold the problem;
// the piece count stays below the kernels' 2^30 limit; the partial-sum sizes do not overflow; the SNP shards are contiguous, aligned and complete.
#include "../../miraculix_amd/csrc/mxa_plan.h"
#include <cstdio>
#include <cstdlib>

using namespace mxa;
static long checks = 0, failures = 0;
#define CHECK(cond, ...) do { checks++; if (!(cond)) { failures++; if (failures < 20) { printf("FAIL %s: ", #cond); printf(__VA_ARGS__); printf("\n"); } } } while (0)

static void check_plan(const GemmPlan &p, long m, long k_pad, int n, bool lut) {
  CHECK(p.splits >= 1, "m %ld k %ld n %d", m, k_pad, n);
  CHECK(p.n_pad >= n && p.m_pad >= m, "m %ld k %ld n %d", m, k_pad, n);
  CHECK((long)p.slabs_total * 128 == k_pad, "m %ld k %ld n %d", m, k_pad, n);
  if (!lut) {
    CHECK(p.a * p.c <= 64 && p.c >= 1 && p.c <= 8 && (p.a == 8 || p.a == 16), "tile a %d c %d", p.a, p.c);
    CHECK(p.n_pad % 4 == 0 && p.n_pad == p.nchunks * 4 * p.c, "n %d n_pad %d", n, p.n_pad);
    CHECK(p.m_pad % (16 * p.a) == 0 && p.m_pad - m < 16 * p.a, "m %ld m_pad %ld", m, p.m_pad);
    // the launcher refuses more than 2^30 - 1 pieces ("launch too large"): never reached by a matrix that fits a 288 GB device with n <= 1024
    if ((double)m * (double)k_pad / 4.0 <= 300e9 && n <= 1024)
      CHECK((long)p.rowblocks * p.nchunks * p.splits <= 0x3fffffffL, "pieces %ld for m %ld k %ld n %d", (long)p.rowblocks * p.nchunks * p.splits, m, k_pad, n);
    const KSplit ks{p.s1, p.l1, p.r1, p.l2};
    long pos = 0;
    for (int sp = 0; sp < p.splits; sp++) {
      const int b = ksplit_begin(ks, sp), l = ksplit_len(ks, sp);
      CHECK(b == pos && l >= 1 && plan_split_begin(p, sp) == b, "split %d of %d begins at %d (expected %ld), length %d; m %ld k %ld n %d", sp, p.splits, b, pos, l, m, k_pad, n);
      pos += l;
    }
    CHECK(pos == p.slabs_total, "splits cover %ld of %d slabs; m %ld k %ld n %d", pos, p.slabs_total, m, k_pad, n);
  } else {
    CHECK((long)p.splits * p.slabs_per_split >= p.slabs_total && (long)(p.splits - 1) * p.slabs_per_split < p.slabs_total, "lut splits %d x %d vs %d", p.splits, p.slabs_per_split, p.slabs_total);
    CHECK(p.m_pad % 512 == 0 && p.m_pad - m < 512, "lut m_pad %ld", p.m_pad);
  }
  const unsigned long long doubles = (unsigned long long)p.splits * (unsigned long long)p.n_pad * (unsigned long long)p.m_pad;
  CHECK(doubles < (1ull << 60), "partial sums %llu doubles: the byte count would overflow", doubles);   // (what does not fit the device runs in split groups: partial_budget)
}

int main() {
  const long ms[] = {1, 3, 127, 128, 129, 500, 1000, 1003, 4096, 50000, 100000, 200000, 625000, 1000000, 2000000, 5000000, 33554432, 40000001};
  const long ks[] = {1, 100, 128, 129, 500, 1000, 12800, 25000, 50000, 100000, 200000, 1000000, 5000000, 40000000};
  const int ns[] = {1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 16, 17, 20, 31, 32, 33, 40, 64, 100, 127, 128, 129, 255, 256, 1000, 65535};
  const long cus_list[] = {256, 64, 304, 1};
  for (long cus : cus_list)
    for (long m : ms)
      for (long k : ks) {
        const long k_pad = (k + 127) / 128 * 128;
        for (int n : ns) {
          const GemmPlan p = plan_gemm_host(m, k_pad, n, cus);
          check_plan(p, m, k_pad, n, false);
          // a row range with the K pieces of the full plan (host-operand pipeline)
          const long mr = std::max<long>(1, m / 3);
          const GemmPlan pr = plan_gemm_host(mr, k_pad, n, cus, &p);
          CHECK(pr.splits == p.splits && pr.s1 == p.s1 && pr.l1 == p.l1 && pr.l2 == p.l2 && pr.r1 == p.r1, "row-range plan differs");
          check_plan(pr, mr, k_pad, n, false);
          if (n <= 4 && cus == 256) check_plan(plan_lut_host(m, k_pad, n), m, k_pad, n, true);
        }
      }
  // SNP shards
  const long snps_list[] = {1, 3, 4, 5, 7, 8, 100, 1000, 1003, 625000, 1000000, 5000000, 2147483647L};
  for (long snps : snps_list)
    for (int want = 1; want <= 64; want++) {
      std::vector<long> b, e;
      shard_blocks(snps, want, b, e);
      CHECK(!b.empty() && (int)b.size() <= want && b.size() == e.size(), "snps %ld want %d: %zu blocks", snps, want, b.size());
      long pos = 0;
      for (size_t g = 0; g < b.size(); g++) {
        CHECK(b[g] == pos && e[g] > b[g] && b[g] % 4 == 0, "snps %ld want %d block %zu = [%ld, %ld), expected begin %ld", snps, want, g, b[g], e[g], pos);
        pos = e[g];
      }
      CHECK(pos == snps, "snps %ld want %d: blocks end at %ld", snps, want, pos);
    }
  printf("plan_sweep: %ld checks, %ld failures\n", checks, failures);
  return failures ? 1 : 0;
}
