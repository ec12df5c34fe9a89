#!/bin/bash
# round 4: the transposed int8 route after the strip / split-plan / finish-kernel changes, against the plain route; n = 3..6 (finish kernel shared with the plain multi-tile route)
cd $GRAFT_REPO_ROOT
for v in 0 1 0 1; do
  export MXA_I8_TN=$v
  echo "== MXA_I8_TN=$v"
  for n in 1 2; do CENTERED=1 timeout -k 10 120 python tools/perf_gemm.py 250000 100000 $n 20 2>&1 | grep tile; done
  timeout -k 10 120 python tools/perf_gram.py 250000 100000 1 2>&1 | grep "gram"
done
unset MXA_I8_TN
echo "== single-orientation object"; MXA_SINGLE_ORIENTATION=1 timeout -k 10 120 python tools/perf_gram.py 250000 100000 1 2>&1 | grep "G\*v"
echo "== two copies, 3 <= n <= 6 on 500k x 50k"
for n in 3 4 5 6; do timeout -k 10 120 python tools/perf_gemm.py 500000 50000 $n 10 2>&1 | grep tile; done
echo "== one copy, 3 <= n <= 6"
for n in 3 4 5 6; do MXA_SINGLE_ORIENTATION=1 timeout -k 10 120 python tools/perf_gemm.py 500000 50000 $n 10 2>&1 | grep "tile.* N "; done
