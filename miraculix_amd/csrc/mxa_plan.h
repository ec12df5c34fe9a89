// mxa_plan.h -- the HOST-ONLY planners of the library: how a product is cut into workgroup tiles, K splits (long pieces + tapered tail) and partial-sum
// buffers (plan_gemm_host, plan_lut_host), and how the SNPs of a multi-device object are cut into shards (shard_blocks).  Plain C++ without any HIP
// dependency, so that exactly this code is compiled by g++ with -fsanitize=address,undefined and swept over shapes on the CPU
// (tests/host/plan_sweep.cpp, `make -C tests/host san`; reference practice: src/miraculix/makefile.c.mk:47-50).  The kernels' geometry constants
// live here for the same reason.
#pragma once
#include <algorithm>
#include <cstddef>
#include <vector>

#if defined(__HIPCC__)
#define MXA_HD __host__ __device__
#else
#define MXA_HD
#endif

namespace mxa {

// geometry of the fp64 MFMA kernel that the planners need
constexpr int kPlanWaves = 4;          // waves per workgroup
constexpr int kPlanSlabSteps = 8;      // K-steps of 16 genotypes per LDS slab
constexpr int kPlanSlabK = 128;        // genotypes per slab
constexpr int kPlanLutWaves = 8;       // lookup kernel: 512 rows per workgroup
constexpr int kPlanLutKS = 128;

// K splits of a launch plan, in units of 128 genotypes (slabs / stages): splits [0, s1) are LONG -- l1 units, the first r1 of them one
// more, so that the long part is covered exactly -- and the rest are SHORT, exactly l2 units each: the TAPERED TAIL.  The pieces of a launch
// are handed out in order, so the last ones are much shorter and the resident slots run dry within a fraction of a long piece's duration.
struct KSplit { int s1, l1, r1, l2; };
MXA_HD inline int ksplit_begin(const KSplit &ks, int sp) {
  return sp < ks.s1 ? sp * ks.l1 + (sp < ks.r1 ? sp : ks.r1) : ks.s1 * ks.l1 + ks.r1 + (sp - ks.s1) * ks.l2;
}
MXA_HD inline int ksplit_len(const KSplit &ks, int sp) { return sp < ks.s1 ? ks.l1 + (sp < ks.r1 ? 1 : 0) : ks.l2; }
// total units, long pieces of about l1_target units, short ones of lt units covering about `tail_units` of the total (0: no tail);
// returns the number of splits
inline int ksplit_make(KSplit &ks, long total, long l1_target, long lt, long tail_units) {
  long s2 = 0;
  if (lt > 0 && tail_units > 0 && 2 * lt <= l1_target) s2 = (tail_units + lt - 1) / lt;
  while (s2 > 0 && s2 * lt > total / 2) s2--;
  const long longpart = total - s2 * lt;
  long s1 = longpart > 0 ? (longpart + l1_target - 1) / l1_target : 0;
  if (s1 < 1 && longpart > 0) s1 = 1;
  const long l1 = s1 ? longpart / s1 : 0;
  ks.s1 = (int)s1; ks.l1 = (int)l1; ks.r1 = (int)(s1 ? longpart - s1 * l1 : 0); ks.l2 = (int)(s2 ? lt : l1);
  return (int)(s1 + s2);
}


// p_rows: rows per tile of the partial-result array P[split][m_pad / p_rows][n_pad][p_rows] (k_gemm: the workgroup's row block, so a
// workgroup writes one contiguous chunk; lookup kernel: m_pad, i.e. plain [split][n_pad][m_pad])
// K splits (mxa_queue.h: KSplit): [0, s1) are l1 slabs long, the first r1 of them one more; [s1, splits) exactly l2 (tapered tail of
// k_gemm; elsewhere s1 = splits, r1 = 0); slabs_per_split = l1
struct GemmPlan { int a, c, nchunks, n_pad, splits, slabs_per_split, slabs_total, rowblocks; long m_pad; long p_rows; int s1, l1, l2, r1; };
inline long plan_split_begin(const GemmPlan &p, int sp) {
  return sp < p.s1 ? (long)sp * p.l1 + (sp < p.r1 ? sp : p.r1) : (long)p.s1 * p.l1 + p.r1 + (long)(sp - p.s1) * p.l2;
}

// resident workgroups of k_gemm per CU by tile: the narrow tiles need few registers (110 / 148 VGPRs for C = 1 / 2 -> 4 / 3 waves per SIMD),
// and more resident waves hide their extraction VALU (n = 4: 55-59 -> 58-64 TFLOP/s); the wide ones fill the register file with 2
inline int gemm_wg_per_cu(int c) { return c == 1 ? 4 : c == 2 ? 3 : 2; }

inline GemmPlan plan_gemm_host(long m, long k_pad, int n, long cus, const GemmPlan *ksplits_like = nullptr) {
  GemmPlan p{};
  // tile choice by n: column chunks of at most 32 columns, C = groups of 4 columns per chunk (balanced over the chunks, so at most
  // 3 padded columns per chunk), A = row groups of 4 per wave: A*C <= 64 accumulators (128 VGPRs)
  p.nchunks = (n + 31) / 32;
  const int per = (n + p.nchunks - 1) / p.nchunks;
  p.c = (per + 3) / 4;
  p.a = p.c <= 4 ? 16 : 8;
  const int cols_chunk = 4 * p.c;
  p.n_pad = p.nchunks * cols_chunk;
  const int rows_wg = kPlanWaves * 4 * p.a;
  p.rowblocks = (int)((m + rows_wg - 1) / rows_wg);
  p.m_pad = (long)p.rowblocks * rows_wg;
  p.p_rows = rows_wg;
  p.slabs_total = (int)(k_pad / kPlanSlabK);
  if (ksplits_like) {   // the same K pieces as another plan (row-range launches of the host-operand pipeline: identical sums)
    p.splits = ksplits_like->splits; p.s1 = ksplits_like->s1; p.l1 = ksplits_like->l1; p.l2 = ksplits_like->l2; p.r1 = ksplits_like->r1; p.slabs_per_split = ksplits_like->slabs_per_split;
    return p;
  }
  // K pieces.  The persistent workgroups (launch_gemm_t) pull pieces = (row block, column chunk, K split) from queues, so what matters is
  // (i) the fixed cost per piece (its epilogue and the turn-around, ~8 us) against its duration and (ii) how the launch ends: the
  // slots run dry over about one piece's duration, half a piece of idle time per slot on average (measured with MXA_DIAG stamps at C2:
  // 0.95-1.0 ms of a 44.6 ms launch with equal pieces).  Hence LONG pieces of ~1.5 ms for the bulk and a TAPERED TAIL: the last
  // ~2.5 rounds' worth of pieces ~0.2 ms long.  Durations from the MFMA count of a slab at the waves per SIMD this tile runs with.
  constexpr double piece_us = 1500.0, tail_us = 200.0;
  const long units = (long)p.rowblocks * p.nchunks;
  const long resident = gemm_wg_per_cu(p.c) * cus;
  const double slab_us = (double)kPlanSlabSteps * p.a * p.c * 16.0 * gemm_wg_per_cu(p.c) / 2390.0;
  long l1 = std::max<long>(8, std::min<long>(p.slabs_total, (long)(piece_us / slab_us + 0.5)));
  // ... and no longer than the K range whose B slabs (C x 4 KiB per slab, streamed by every piece of a group) stay in one XCD's 4 MiB L2 next to
  // the packed rows passing through: 3 MiB.  Measured at C2 (rocprofv3 FETCH_SIZE x 2 + WRITE_SIZE per launch): pieces of 1.5 ms (214 / 181 slabs,
  // 6.8 / 5.8 MB of B) 25.3 GB, pieces of 0.66 ms (96 slabs, 3 MB) 18.9 GB against 12.8 GB algorithmic, for 0.3 % of the time.
  constexpr long b_l2_bytes = 3L << 20;
  {
    const long cap = b_l2_bytes / ((long)p.c * 4096);
    // (relaxing the cap for a short K -- C2 'T' in 2-3 splits instead of 5 -- gained 0.3-0.5 % of that call for 6.5 GB more fabric traffic: removed)
    l1 = std::max<long>(8, std::min<long>(l1, cap));
  }
  while (l1 > 16 && units * ((p.slabs_total + l1 - 1) / l1) < 6 * resident) l1 = l1 * 3 / 4;   // at least ~6 rounds of pieces: the queues balance the slots
  const long lt = tail_us > 0 ? std::max<long>(8, (long)(tail_us / slab_us + 0.5)) : 0;
  const bool taper = lt > 0 && units * ((p.slabs_total + l1 - 1) / l1) >= 3 * resident;
  const long tail = taper ? lt * std::max<long>(1, (5 * resident / 2 + units - 1) / units) : 0;   // short splits: about 2.5 rounds of short pieces
  KSplit ks;
  p.splits = ksplit_make(ks, p.slabs_total, l1, lt, tail);
  p.s1 = ks.s1; p.l1 = ks.l1; p.r1 = ks.r1; p.l2 = ks.l2;
  l1 = ks.l1;
  p.slabs_per_split = (int)l1;
  if (p.splits < 1) p.splits = 1;
  return p;
}


inline GemmPlan plan_lut_host(long m, long k_pad, int n) {
  GemmPlan p{};
  p.a = 0; p.c = 0;
  p.n_pad = n <= 1 ? 1 : n <= 2 ? 2 : 4;
  p.nchunks = 1;
  p.rowblocks = (int)((m + 64 * kPlanLutWaves - 1) / (64 * kPlanLutWaves));
  p.m_pad = (long)p.rowblocks * 64 * kPlanLutWaves;
  p.p_rows = p.m_pad;
  p.slabs_total = (int)(k_pad / kPlanLutKS);
  const long units = p.rowblocks;
  constexpr long target = 8192L;
  long want = (target + units - 1) / units;
  long max_splits = std::max<long>(1, p.slabs_total / 16);
  long splits = std::max<long>(1, std::min<long>(want, max_splits));
  p.slabs_per_split = (int)((p.slabs_total + splits - 1) / splits);
  p.splits = (p.slabs_total + p.slabs_per_split - 1) / p.slabs_per_split;
  p.s1 = p.splits; p.l1 = p.l2 = p.slabs_per_split; p.r1 = 0;
  return p;
}


inline void shard_blocks(long snps, int want, std::vector<long> &b, std::vector<long> &e) {
  // contiguous blocks at multiples of 4, the same rule as miraculix_amd/distributed.py:shard_bounds; shards that
  // would be empty (4 * shards > snps) are dropped
  const long per = ((snps + want - 1) / want + 3) / 4 * 4;
  for (int g = 0; g < want; g++) {
    const long b0 = std::min(snps, g * per), e0 = std::min(snps, b0 + per);
    if (e0 > b0) { b.push_back(b0); e.push_back(e0); }
  }
}


}  // namespace mxa
