#!/bin/bash
# round 4: half-tile workgroups of k_gemm_i8 (n <= 2): MXA_I8_HALF_TILE=0 (whole tiles + K splits), =1 (always half tiles), unset (the plan's cost model decides)
cd $GRAFT_REPO_ROOT
for ht in 0 1 ""; do
  [ -n "$ht" ] && export MXA_I8_HALF_TILE=$ht || unset MXA_I8_HALF_TILE
  echo "== MXA_I8_HALF_TILE=${ht:-unset}"
  for n in 1 2; do CENTERED=1 timeout -k 10 120 python tools/perf_gemm.py 250000 100000 $n 20 2>&1 | grep tile; done
  timeout -k 10 120 python tools/perf_gram.py 250000 100000 1 2>&1 | grep "G\*v"
  CENTERED=1 timeout -k 10 120 python tools/perf_gemm.py 2000000 100000 1 5 2>&1 | grep tile
  CENTERED=1 timeout -k 10 120 python tools/perf_gemm.py 60000 30000 1 20 2>&1 | grep tile
done
