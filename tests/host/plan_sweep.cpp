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

// schedule of k_gemm_i8_tn: run every workgroup's item list and require that each strip's K range is covered exactly once, that a strip's pieces fill the
// slots 0 .. tn_pieces(strip) - 1 exactly once each (k_finish_i8_t adds exactly those), and that no piece outgrows the int32 accumulators
static void check_tn(long indiv, long snps, long slots, int tiles, int force) {
  const long slabs = (indiv + 127) / 128;
  const TnSched s = plan_i8_tn_host(slabs, snps, slots, tiles, force);
  const long n = (slabs + 1) / 2, K = (snps + 255) / 256;
  CHECK(s.strips == n && s.K == K && s.slots == slots && (force < 0 || s.mode == force || (force == 1 && s.mode == 0)), "indiv %ld snps %ld slots %ld: header", indiv, snps, slots);
  CHECK(s.pslots >= 1 && s.pslots <= kTnMaxPieces, "indiv %ld snps %ld slots %ld mode %d: %d slots of P", indiv, snps, slots, s.mode, s.pslots);
  std::vector<int> covered((size_t)n * K, 0), slot_used((size_t)n * kTnMaxPieces, 0);
  const int grid = tn_grid(s);
  CHECK(grid >= 1 && grid <= slots, "grid %d", grid);
  long pieces = 0;
  bool too_long = false;
  for (int wg = 0; wg < grid; wg++) {
    int i = 0, g = 0; TnItem it;
    int guard = 0;
    while (tn_next(s, wg, i, g, it)) {
      if (++guard > 100000) { CHECK(false, "workgroup %d never ends", wg); break; }
      pieces++;
      const bool in = it.strip >= 0 && it.strip < n && it.st0 >= 0 && it.stages >= 1 && it.st0 + it.stages <= K && it.slot >= 0 && it.slot < s.pslots;
      CHECK(in, "indiv %ld snps %ld slots %ld mode %d wg %d: item strip %d [%d, +%d) slot %d", indiv, snps, slots, s.mode, wg, it.strip, it.st0, it.stages, it.slot);
      if (!in) continue;
      if (it.stages > kTnMaxPieceStages) too_long = true;
      for (int q = 0; q < it.stages; q++) covered[(size_t)it.strip * K + it.st0 + q]++;
      slot_used[(size_t)it.strip * kTnMaxPieces + it.slot]++;
    }
  }
  // (pieces beyond 2047 stages only where the caller refuses the product: K > c_lo x 2047 in mode 0)
  CHECK(!too_long || (s.mode == 0 && (K + s.q1 - 1) / s.q1 > kTnMaxPieceStages), "indiv %ld snps %ld: a piece outgrows the accumulators", indiv, snps);
  long bad_cover = 0, bad_slots = 0;
  for (size_t q = 0; q < covered.size(); q++) bad_cover += covered[q] != 1;
  for (long st = 0; st < n; st++) {
    const int pc = tn_pieces(s, st);
    if (pc < 1 || pc > s.pslots) { bad_slots++; continue; }
    for (int q = 0; q < kTnMaxPieces; q++) bad_slots += slot_used[(size_t)st * kTnMaxPieces + q] != (q < pc ? 1 : 0);
  }
  CHECK(bad_cover == 0, "indiv %ld snps %ld slots %ld tiles %d mode %d: %ld stages not covered exactly once", indiv, snps, slots, tiles, s.mode, bad_cover);
  CHECK(bad_slots == 0, "indiv %ld snps %ld slots %ld tiles %d mode %d: %ld slot mismatches", indiv, snps, slots, tiles, s.mode, bad_slots);
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
  // schedules of the transposed-operand int8 kernel, planner's choice and both modes forced
  const long tn_indiv[] = {1, 200, 256, 257, 1301, 10000, 50000, 65536, 100000, 131072, 133000, 200000, 262144, 500000};
  const long tn_snps[] = {1, 255, 256, 777, 6000, 10000, 50000, 250000, 524032, 524033, 1000000};
  const long tn_slots[] = {512, 256, 208, 608, 8};
  for (long slots : tn_slots)
    for (long indiv : tn_indiv)
      for (long snps : tn_snps) {
        if ((indiv / 256 + 1) * (snps / 256 + 1) > 6000000) continue;
        for (int force = -1; force <= 1; force++) check_tn(indiv, snps, slots, slots >= 500 ? 1 : 2, force);
      }
  {   // random shapes, slot counts up to 2048 (the 32-bit decode of mode 1 is admitted by the planner only while nb Q < 2^32: shapes near that edge are in here)
    unsigned long long x = 88172645463325252ull;
    auto rnd = [&](unsigned long long m) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return (long)(x % m); };
    for (int it = 0; it < 1500; it++) {
      const long slots = 1 + rnd(2048), strips = 1 + rnd(it % 3 ? 3000 : 60), K = 1 + rnd(it % 5 ? 2047 : 6000);
      if (strips * K > 4000000) continue;
      const long indiv = strips * 256 - rnd(256), snps = K * 256 - rnd(256);
      for (int force = -1; force <= 1; force++) check_tn(std::max(1L, indiv), std::max(1L, snps), slots, 1 + (int)rnd(2), force);
    }
  }
  {   // the headline shard takes the two-class schedule: 391 heads + 121 tail workgroups, three slots of P
    const TnSched s = plan_i8_tn_host((100000 + 127) / 128, 250000, 512, 1);
    CHECK(s.mode == 1 && s.q0 == 0 && s.q1 == 391 && s.q2 == 761 && s.pslots == 3, "headline schedule: mode %d whole %d rem %d la %d pslots %d", s.mode, s.q0, s.q1, s.q2, s.pslots);
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
