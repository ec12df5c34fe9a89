#!/bin/bash
# the gated fp64 fallback behind guarded chains of 2..6 columns: tests, then the price when the int8 classes apply and the time when they do not (500k x 50k)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05b; O=$GRAFT_REPO_ROOT/gpurun_out/r05b
timeout -k 10 900 python -m pytest tests/test_small_n_gpu.py tests/test_single_orientation_gpu.py tests/test_async_gpu.py tests/test_engine_i8_guarded_gpu.py tests/test_edge_gpu.py tests/test_dgemm_gpu.py tests/test_grouped_and_incremental_gpu.py -q -x -p no:cacheprovider > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -2 $O/tests.log
: > $O/perf.txt
for so in 1 0; do for n in 2 4 6; do
  echo "== copies: $((2-so)) n=$n exact" >> $O/perf.txt
  MXA_SINGLE_ORIENTATION=$so CENTERED=1 timeout -k 10 200 python3 tools/perf_gemm.py 500000 50000 $n 20 2>&1 | grep tile >> $O/perf.txt || exit 1
  echo "== copies: $((2-so)) n=$n one entry 150 binades down (class 2)" >> $O/perf.txt
  MXA_SINGLE_ORIENTATION=$so TINY_ENTRY=1 CENTERED=1 timeout -k 10 200 python3 tools/perf_gemm.py 500000 50000 $n 5 2>&1 | grep tile >> $O/perf.txt || exit 1
done; done
cut -c1-170 $O/perf.txt
