#!/bin/bash
# n = 1..8 on the config-5 shard: lookup kernel vs MFMA tiles (MXA_LUT_MAX_N) vs the opt-in int8 engine
cd "$GRAFT_REPO_ROOT"
for n in 1 2 3 4 6 8; do
  echo "--- n=$n default"; CENTERED=1 timeout -k 10 200 python tools/perf_gemm.py 250000 100000 $n 5 2>&1 | grep -v amdgpu
  if [ $n -le 4 ]; then echo "--- n=$n all-MFMA"; CENTERED=1 MXA_LUT_MAX_N=0 timeout -k 10 200 python tools/perf_gemm.py 250000 100000 $n 5 2>&1 | grep -v amdgpu; fi
  if [ $n -le 4 ]; then echo "--- n=$n LUT"; CENTERED=1 MXA_LUT_MAX_N=4 timeout -k 10 200 python tools/perf_gemm.py 250000 100000 $n 5 2>&1 | grep -v amdgpu; fi
  echo "--- n=$n i8"; CENTERED=1 MXA_ENGINE=i8 timeout -k 10 200 python tools/perf_gemm.py 250000 100000 $n 5 2>&1 | grep -v amdgpu
done
