#!/bin/bash
# A/B of the transposed-operand k_gemm (MXA_GEMM_TR=1: each product from the copy whose rows are the K index) against the plain form, and of its two
# conversion variants (MXA_GEMM_TR_MODE 2 = v_bfe_u32, 3 = v_and_b32 with the row scale undone in the epilogue); same box, alternating
set -e
for shape in "1000000 50000 32 6" "500000 50000 8 6" "500000 50000 16 6" "500000 50000 12 6" "500000 50000 20 6" "500000 50000 64 4" "625000 200000 128 2"; do
  for v in "0 2" "1 2" "1 3" "0 2" "1 2" "1 3"; do
    set -- $v
    echo "MXA_GEMM_TR=$1 MXA_GEMM_TR_MODE=$2"; MXA_GEMM_TR=$1 MXA_GEMM_TR_MODE=$2 python tools/perf_gemm.py $shape 2>&1 | grep "tile="
  done
done
