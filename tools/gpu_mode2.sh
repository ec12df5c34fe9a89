#!/bin/bash
set -e
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_dgemm_gpu.py tests/test_edge_gpu.py tests/test_cg_gpu.py tests/test_shard_gpu.py tests/test_staging_gpu.py -x -q > gpurun_out/m2_test.log 2>&1 || { tail -40 gpurun_out/m2_test.log; exit 1; }
tail -2 gpurun_out/m2_test.log
for m in 0 2; do
MXA_GEMM_MODE=$m timeout -k 10 300 python tools/perf_gemm.py 1000000 50000 32 5 2>&1 | grep -v amdgpu
done
MXA_DIAG=1 MXA_GEMM_MODE=2 timeout -k 10 300 python tools/perf_gemm.py 1000000 50000 32 1 2>&1 | grep DIAG | sort -u | head -3
for n in 10 20 128; do timeout -k 10 300 python tools/perf_gemm.py 500000 50000 $n 3 2>&1 | grep -v amdgpu; done
