#!/bin/bash
# round 4: 'N' products with 3 <= n <= 6 (and n = 9, 10: peeled columns) on a single-orientation object -- k_gemm_i8_tn, one pass per tile of 32 expanded columns -- against a two-copy object
cd $GRAFT_REPO_ROOT
for so in 0 1; do
  export MXA_SINGLE_ORIENTATION=$so
  echo "== MXA_SINGLE_ORIENTATION=$so"
  for n in 3 4 5 6 9 10; do timeout -k 10 120 python tools/perf_gemm.py 500000 50000 $n 10 2>&1 | grep "tile.* N "; done
done
