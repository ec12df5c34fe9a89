#!/bin/bash
# round 5: k_gemm with the permuted K order.  Parity first (plain == transposed bit for bit, oracle), then A/B: MXA_GEMM_TR=0 (plain form) against
# MXA_GEMM_TR=1 (transposed-operand form), conversion variants MXA_GEMM_MODE 2 (v_bfe_u32) / 3 (v_and_b32); same box, alternating
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gemm_tr_gpu.py tests/test_dgemm_gpu.py tests/test_edge_gpu.py tests/test_single_orientation_gpu.py tests/test_host_pipeline_gpu.py -x -q -m gpu > gpurun_out/r5_plain_tests.log 2>&1 || { tail -30 gpurun_out/r5_plain_tests.log; exit 1; }
tail -2 gpurun_out/r5_plain_tests.log
{
for shape in "1000000 50000 32 6" "500000 50000 8 6" "500000 50000 16 6" "500000 50000 20 6" "625000 200000 128 2"; do
  for v in "0 3" "0 2" "1 3" "0 3" "0 2" "1 3"; do
    set -- $v
    echo "MXA_GEMM_TR=$1 MXA_GEMM_MODE=$2"; MXA_GEMM_TR=$1 MXA_GEMM_MODE=$2 MXA_GEMM_TR_MODE=$2 timeout -k 10 300 python tools/perf_gemm.py $shape 2>&1 | grep "tile="
  done
done
} > gpurun_out/r5_plain_ab.txt 2>&1
cat gpurun_out/r5_plain_ab.txt
