#!/bin/bash
# A/B of the transposed-operand exact int8 route for n <= 2 (MXA_I8_TN=1: k_gemm_i8_tn, each product from the copy whose rows are the K index) against the
# plain k_gemm_i8<1,2,1> on the config-5 shard; then the split counts of the transposed kernel
for v in 0 1 0 1; do
  echo "MXA_I8_TN=$v"; for n in 1 2; do MXA_I8_TN=$v CENTERED=1 python tools/perf_gemm.py 250000 100000 $n 20 2>&1 | grep tile; done
  MXA_I8_TN=$v python tools/perf_gram.py 250000 100000 1 2>&1 | grep "G\*v"
done
for sp in 4 5 9 13 18 23; do echo "MXA_I8_TN=1 MXA_I8_TN_SPLITS=$sp"; MXA_I8_TN=1 MXA_I8_TN_SPLITS=$sp CENTERED=1 python tools/perf_gemm.py 250000 100000 1 20 2>&1 | grep tile; done
