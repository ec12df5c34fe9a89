#!/bin/bash
# round 5: tapered K splits of the int8 kernels (MXA_I8_TAPER: 0 = uniform lengths, 0.4 = lengths from 1.4x to 0.6x of the mean, dispatched longest first), two copies / one copy
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_async_gpu.py tests/test_small_n_gpu.py tests/test_i8_tn_gpu.py tests/test_single_orientation_gpu.py tests/test_cg_gpu.py tests/test_engine_i8_gpu.py -x -q -m gpu > gpurun_out/r5_taper_tests.log 2>&1 || { tail -30 gpurun_out/r5_taper_tests.log; exit 1; }
tail -2 gpurun_out/r5_taper_tests.log
{
for rep in 1 2; do
for tp in 0 0.25 0.4 0.6; do
for so in 0 1; do
  echo "== MXA_I8_TAPER=$tp MXA_SINGLE_ORIENTATION=$so"
  MXA_I8_TAPER=$tp MXA_SINGLE_ORIENTATION=$so CENTERED=1 timeout -k 10 200 python tools/perf_gemm.py 250000 100000 1 20 2>&1 | grep "tile="
  MXA_I8_TAPER=$tp MXA_SINGLE_ORIENTATION=$so timeout -k 10 200 python tools/perf_gram.py 250000 100000 1 2>&1 | grep "gram_matvec"
done
done
done
for tp in 0 0.4; do
  for n in 4 6; do
    echo "== MXA_I8_TAPER=$tp two copies"
    MXA_I8_TAPER=$tp MXA_SINGLE_ORIENTATION=0 timeout -k 10 200 python tools/perf_gemm.py 500000 50000 $n 10 2>&1 | grep "tile="
  done
done
} > gpurun_out/r5_taper_ab.txt 2>&1
cat gpurun_out/r5_taper_ab.txt
