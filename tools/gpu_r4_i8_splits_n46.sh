#!/bin/bash
# round 4: K splits of the exact int8 route at 3 <= n <= 6 (two digit tiles): the plan's choice against forced counts, 500k x 50k
cd $GRAFT_REPO_ROOT
for sp in "" 1 2 3 5 8 11; do
  [ -n "$sp" ] && export MXA_I8_SPLITS=$sp || unset MXA_I8_SPLITS
  echo "== MXA_I8_SPLITS=${sp:-plan}"
  for n in 4 6; do timeout -k 10 120 python tools/perf_gemm.py 500000 50000 $n 10 2>&1 | grep tile; done
done
