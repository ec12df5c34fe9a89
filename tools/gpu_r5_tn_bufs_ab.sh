#!/bin/bash
# round 5: k_gemm_i8_tn with the digit fragments in registers: 4 buffers x 2 workgroups per CU (3 stages in flight) against 3 x 3 (2 in flight); one-copy object, config-5 shard
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_i8_tn_gpu.py tests/test_single_orientation_gpu.py tests/test_async_gpu.py -x -q -m gpu > gpurun_out/r5_tn_tests.log 2>&1 || { tail -30 gpurun_out/r5_tn_tests.log; exit 1; }
tail -2 gpurun_out/r5_tn_tests.log
{
for rep in 1 2; do
for b in 4; do
  echo "== MXA_I8_TN_BUFS=$b one copy"
  MXA_I8_TN_BUFS=$b MXA_SINGLE_ORIENTATION=1 CENTERED=1 timeout -k 10 200 python tools/perf_gemm.py 250000 100000 1 20 2>&1 | grep "tile="
  MXA_I8_TN_BUFS=$b MXA_SINGLE_ORIENTATION=1 timeout -k 10 200 python tools/perf_gram.py 250000 100000 1 2>&1 | grep "gram_matvec"
done
echo "== two copies"
MXA_SINGLE_ORIENTATION=0 CENTERED=1 timeout -k 10 200 python tools/perf_gemm.py 250000 100000 1 20 2>&1 | grep "tile="
MXA_SINGLE_ORIENTATION=0 timeout -k 10 200 python tools/perf_gram.py 250000 100000 1 2>&1 | grep "gram_matvec"
done
for n in 3 4 6; do
  echo "== one copy n=$n"
  MXA_SINGLE_ORIENTATION=1 timeout -k 10 200 python tools/perf_gemm.py 500000 50000 $n 10 2>&1 | grep "tile="
done
} > gpurun_out/r5_tn_bufs_ab.txt 2>&1
cat gpurun_out/r5_tn_bufs_ab.txt
