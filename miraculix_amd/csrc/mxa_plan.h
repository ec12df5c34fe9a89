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


// ---- schedule of the transposed-operand int8 kernel (k_gemm_i8_tn, mxa_gemm_i8.hip): which pieces of which strips a persistent workgroup multiplies.
// A STRIP = 256 individuals (2 slabs), its K range = `K` stages of 256 SNP rows; a PIECE = a stage range of one strip whose int32 sums go to one slot of
// P[slot][e][individual]; k_finish_i8_t adds a strip's slots exactly (int64).  Integer sums do not depend on the cuts, so the cuts are free to serve the clock:
// every piece costs its stages plus about kTnPieceCost stages of start-up, LDS reduction and 32 KiB of partial sums per digit tile leaving for HBM (round 6: those
// stores, 72 MB per product at 5.6 pieces per strip, were 8 % of the kernel -- profiles/r06_tn_ablation.txt).
//  mode 0 (round 5): every strip in c or c + 1 EQUAL pieces so that the items fill whole rounds of the resident slots; the items of a round sweep their K ranges
//    in step, the digit fragments they share stay in the L2s.
//  mode 1 (round 6), TWO CLASSES: every workgroup first takes `whole` whole strips (slot 0, in step).  Of the `rem` < slots strips left, workgroup w < rem takes
//    the HEAD [0, la) of strip w (slot 0); the other nb = slots - rem workgroups share the TAILS [la, K) of those strips evenly as one strip-major sequence of
//    rem * (K - la) stages (a tail piece: slot 1 + its workgroup's distance from the first workgroup of that tail).  rem + slots pieces instead of ~ rounds * slots:
//    250 000 x 100 000 on 512 slots = 391 heads of 761 stages + 512 tail pieces (2.3 per strip, 3 slots of P) where mode 0 cut 2 048 pieces (5.2 per strip, 6 slots).
//    The tail workgroups start at different K positions -- they keep cycling over the tails' K window, which (with two workgroups per CU) has to stay
//    L2-resident.  When mode 1 is taken: plan_i8_tn_host, from measurements.
constexpr int kTnPieceCost = 20;
constexpr int kTnMaxPieces = 64;                    // partial-sum slots per strip at most
constexpr int kTnMaxPieceStages = 2047;             // int32 accumulators
constexpr long kTnTailWindowBytes = 3L << 20;      // one tile per pass: the tails' digit fragments (8 KiB per stage) have to stay L2-resident
constexpr int kTnTailMinStages = 96;
// (q0..q2 by mode, so that the kernel holds seven scalars: 0 -- items, c_lo, n_lo: strips [0, n_lo) in c_lo pieces, the others in c_lo + 1;  1 -- whole, rem, la)
struct TnSched { int mode, strips, K, slots, pslots, q0, q1, q2; };   // pslots = slots of P (the largest piece count of a strip)
struct TnItem { int strip, st0, stages, slot; };
// mode 1, tails: nb = slots - rem workgroups share Q = rem (K - la) stages (< 2^31: rem < slots, K <= 2047); workgroup b begins at floor(b Q / nb)
// (32-bit arithmetic -- the kernel decodes with it between two items, with all its registers taken: the planner admits mode 1 only while nb Q < 2^32)
MXA_HD inline int tn_tail_begin(int Q, int nb, int b) { return (int)((unsigned)b * (unsigned)Q / (unsigned)nb); }
MXA_HD inline int tn_tail_owner(int Q, int nb, int g) { return (int)((((unsigned)g + 1u) * (unsigned)nb - 1u) / (unsigned)Q); }   // the tail workgroup whose range holds stage g
// workgroups launched (mode 1 needs every one of its slots)
MXA_HD inline int tn_grid(const TnSched &s) { return s.mode == 0 && s.q0 < s.slots ? s.q0 : s.slots; }
// pieces of a strip = slots the finish kernel adds
MXA_HD inline int tn_pieces(const TnSched &s, long strip) {
  if (s.mode == 0) return s.q1 + (strip >= s.q2 ? 1 : 0);
  const long j = strip - (long)s.q0 * s.slots;
  const int rem = s.q1, tail = s.K - s.q2;
  if (j < 0 || tail <= 0) return 1;
  return 2 + tn_tail_owner(rem * tail, s.slots - rem, (int)(j + 1) * tail - 1) - tn_tail_owner(rem * tail, s.slots - rem, (int)j * tail);
}
// the i-th item of workgroup wg (i counts from 0; g is the workgroup's cursor in the tail sequence, carried between calls); false: no more items
MXA_HD inline bool tn_next(const TnSched &s, int wg, int &i, int &g, TnItem &it) {
  if (s.mode == 0) {
    const int nitems = s.q0, c_lo = s.q1, n_lo = s.q2;
    const int item = wg + i * s.slots;               // (items < 2^31: the planner's rounds are bounded)
    if (item >= nitems) return false;
    int c, piece;
    if (item < n_lo * c_lo) { c = c_lo; piece = item / n_lo; it.strip = item - piece * n_lo; }     // piece-major: the strips' p-th pieces are neighbours
    else { const int j = item - n_lo * c_lo, n_hi = s.strips - n_lo; c = c_lo + 1; piece = j / n_hi; it.strip = n_lo + j - piece * n_hi; }
    it.st0 = (int)((unsigned)piece * (unsigned)s.K / (unsigned)c);            // piece < 64, K < 2^25
    it.stages = (int)((unsigned)(piece + 1) * (unsigned)s.K / (unsigned)c) - it.st0;
    it.slot = piece;
    i++;
    return true;
  }
  const int whole = s.q0, rem = s.q1, la = s.q2;
  if (i < whole) { it.strip = i * s.slots + wg; it.st0 = 0; it.stages = s.K; it.slot = 0; i++; return true; }
  if (rem == 0) return false;
  const int base = whole * s.slots;
  if (wg < rem) {
    if (i > whole) return false;
    it.strip = base + wg; it.st0 = 0; it.stages = la; it.slot = 0; i++;
    return true;
  }
  const int tail = s.K - la;
  if (tail <= 0) return false;
  const int b = wg - rem, nb = s.slots - rem, Q = rem * tail;
  if (i == whole) g = tn_tail_begin(Q, nb, b);
  const int ge = tn_tail_begin(Q, nb, b + 1);
  if (g >= ge) return false;
  const int j = g / tail, off = g - j * tail;
  const int len = ge - g < tail - off ? ge - g : tail - off;
  it.strip = base + j; it.st0 = la + off; it.stages = len; it.slot = 1 + b - tn_tail_owner(Q, nb, j * tail);
  g += len; i++;
  return true;
}
// indiv_slabs x snp_rows = the SNP-major copy; slots = resident workgroups of the instantiation (CUs x 2 for one digit tile per pass, x 1 for two);
// force_mode >= 0: tests and A/B runs.  K beyond c x 2047 stages is the caller's to refuse (pieces longer than kTnMaxPieceStages: mode 0 with q1 = c_lo at its minimum).
inline TnSched plan_i8_tn_host(long indiv_slabs, long snp_rows, long slots, int tiles_per_pass, int force_mode = -1) {
  TnSched s{};
  const int n = (int)((indiv_slabs + 1) / 2), K = (int)((snp_rows + 255) / 256);
  s.strips = n; s.K = K; s.slots = (int)slots;
  // mode 0: R rounds of equal pieces
  const int c_min = std::max(1, (K + kTnMaxPieceStages - 1) / kTnMaxPieceStages);
  const int c_max = std::max(c_min, std::min(kTnMaxPieces - 1, K / 24));   // pieces shorter than ~24 stages are mostly start-up
  double best = -1.0; int best_c = c_min, best_nlo = n;
  for (long R = 1; R <= 64; R++) {
    const long T = R * slots;
    int c = (int)std::min<long>(c_max, std::max<long>(c_min, T / n));
    long n_hi = std::min<long>(n, std::max<long>(0, T - (long)n * c));
    if (c >= c_max) n_hi = 0;
    const long n_lo = n - n_hi, items_lo = n_lo * c, items = items_lo + n_hi * (c + 1);
    const double len_lo = (double)K / c, len_hi = (double)K / (c + 1);
    double cost = 0.0;
    for (long i0 = 0; i0 < items; i0 += slots) cost += (i0 < items_lo ? len_lo : len_hi) + kTnPieceCost;   // a round lasts as long as its first (longest) item + start-up and flush (fits the measured shapes)
    cost += 0.35 * (double)(items) / (double)n;                                                             // the finish kernel reads one slot per piece
    if (best < 0 || cost < best * 0.995) { best = cost; best_c = c; best_nlo = (int)n_lo; }
    if (c >= c_max) break;
  }
  s.mode = 0; s.q0 = best_nlo * best_c + (n - best_nlo) * (best_c + 1); s.q1 = best_c; s.q2 = best_nlo;
  s.pslots = best_c + (best_nlo < n ? 1 : 0);
  // mode 1: whole strips, then heads and tails
  if (force_mode != 0 && K <= kTnMaxPieceStages && n > 0) {
    TnSched t = s;
    const int whole = (int)(n / slots), rem = (int)(n - (long)whole * slots), nb = rem ? (int)slots - rem : 0;
    t.mode = 1; t.q0 = whole; t.q1 = rem; t.q2 = K; t.pslots = 1;
    bool valid = true, liked = true;                   // valid: a schedule the kernels can run; liked: one the planner takes unforced
    double cost = (double)whole * (K + kTnPieceCost), pieces = (double)whole * (double)slots;
    if (rem) {
      // heads and tail shares end together: la + cost = rem (K - la) / nb + cost (1 + rem / nb)
      const long la = std::max<long>(1, std::min<long>(K, ((long)rem * (K + kTnPieceCost) + slots / 2) / slots));
      t.q2 = (int)la;
      const long tail = K - la, Q = (long)rem * tail;
      if (tail > 0) {
        valid = Q >= nb && (double)Q * (double)nb < 4294967296.0;   // every tail workgroup has at least one stage; 32-bit decode
        if (valid) {
          int most = 0;
          for (long j = 0; j < rem; j++) most = std::max(most, tn_pieces(t, (long)whole * slots + j));
          t.pslots = most;
          valid = most <= kTnMaxPieces;
        }
        // measured, both modes alternating inside one process (tools/ab_env.py; profiles/r06_tn_sched_ab.txt).  Two digit tiles per pass (one workgroup per CU,
        // nobody hides a piece's start-up and flush): mode 1 won or tied on twelve of thirteen shapes (0-16 %; 250 000 x 100 000, n = 4: 1.49 -> 1.44 ms) and lost
        // 1.6 % on the one where under a third of the slots hold heads (79 strips).  One tile per pass (two workgroups per CU): +0.5-2.4 % while the tails are
        // 100-370 stages long and a tail workgroup has at most ~5 of them; it LOST 1-8 % with short tails in many pieces (120 000 individuals: 64 stages, 12 pieces
        // per workgroup; K = 123: 14 stages), with tails whose digit window outgrows the L2 (447 and 600 stages) and where most workgroups are tail workgroups.
        liked = Q >= (long)nb * 24 && la >= 24 && t.pslots <= 8 &&
                (tiles_per_pass >= 2 ? 2L * rem >= nb : (tail >= kTnTailMinStages && tail * 8192L <= kTnTailWindowBytes && rem >= nb && rem <= 5L * nb));
        cost += std::max<double>((double)la + kTnPieceCost, (double)Q / nb + kTnPieceCost * (1.0 + (double)rem / nb));
        pieces += 2.0 * rem + nb;
      } else { cost += (double)la + kTnPieceCost; pieces += rem; }
    }
    cost += 0.35 * pieces / (double)n;
    if (valid && (force_mode == 1 || (liked && (tiles_per_pass >= 2 || cost < best)))) s = t;
  }
  return s;
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
