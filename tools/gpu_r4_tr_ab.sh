#!/bin/bash
# A/B of the transposed-operand k_gemm (MXA_GEMM_TR=1: 'N' from the SNP-major copy) against the launch on the individual-major copy, same box, same process order
set -e
for shape in "1000000 50000 32 6" "625000 200000 128 2" "500000 50000 8 6" "500000 50000 16 6" "500000 50000 64 4"; do
  for tr in 0 1 0 1; do
    echo "MXA_GEMM_TR=$tr"; MXA_GEMM_TR=$tr python tools/perf_gemm.py $shape | grep " N "
  done
done
