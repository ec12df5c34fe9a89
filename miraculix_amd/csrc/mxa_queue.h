// mxa_queue.h -- work queues of the persistent kernels (k_gemm, k_gemm_i8): device side.
//
// A launch has one workgroup per resident slot; each pulls pieces (row block, column chunk, K split) until the queues are empty.  Pieces are
// numbered like the block index of a one-piece-per-workgroup launch: row block fastest inside a (column chunk, K split) GROUP, whose
// workgroups stream the same B slabs.  The first g8 groups (a multiple of 8, long pieces only) are dealt whole to the 8 XCDs -- each has its
// own L2, so a group's B slabs cross the fabric once instead of once per XCD: queue x (counter x) holds the pieces of XCD x in order;
// queue 8 holds the remaining groups (the rest of the long ones, then the short pieces of the tapered tail) in plain order for everybody; a
// workgroup whose queues are empty steals from the other XCDs' queues.  The XCD of a workgroup is read from the hardware, not assumed.
// ctr: 9 counters zeroed by the launcher on the same stream.  A fetch is one returning agent-scope atomic add by ONE lane of the workgroup;
// the caller broadcasts the result through LDS.  Every workgroup leaves when a fetch returns -1: the grid always drains.
#pragma once
#include <hip/hip_runtime.h>
#include "mxa_plan.h"   // KSplit

namespace mxa {

__device__ __forceinline__ int hw_xcc_id() {
  int x;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(x));
  return x & 7;
}

struct PieceQueue {
  int *ctr;
  int rowblocks, g8, n1, n2, xcc, phase;   // n1: pieces in every XCD queue; n2: pieces in the common queue
  __device__ __forceinline__ void init(int *counters, int rowblocks_, int g8_, int nunits) {
    ctr = counters; rowblocks = rowblocks_; g8 = g8_;
    n1 = (g8 / 8) * rowblocks;
    n2 = nunits - g8 * rowblocks;
    xcc = hw_xcc_id();
    phase = 0;                               // 0 own XCD queue, 1 common queue, 2..8 the other XCDs' queues
  }
  // next piece number, or -1
  __device__ __forceinline__ int next() {
    while (phase <= 8) {
      if (phase == 1) {
        const int t = __hip_atomic_fetch_add(ctr + 8, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (t < n2) return g8 * rowblocks + t;
      } else {
        const int y = phase == 0 ? xcc : ((xcc + phase - 1) & 7);
        const int sl = n1 > 0 ? __hip_atomic_fetch_add(ctr + y, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : n1;
        if (sl < n1) return sl * 8 + y;
      }
      phase++;
    }
    return -1;
  }
  // piece number -> (row block, group)
  __device__ __forceinline__ void locate(int bid, int &rb, int &grp) const {
    if (bid < g8 * rowblocks) { const int xcd = bid & 7, slot = bid >> 3; rb = slot % rowblocks; grp = xcd + 8 * (slot / rowblocks); }
    else { const int t = bid - g8 * rowblocks; rb = t % rowblocks; grp = g8 + t / rowblocks; }
  }
};

}  // namespace mxa
