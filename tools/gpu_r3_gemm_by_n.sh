#!/bin/bash
# round 3: dgemm_compressed by n on 500k x 50k under a warm clock.  Default engine: 3 <= n <= 6 on the exact int8 route (digits from the span of B),
# odd columns of n > 6 peeled into it, the rest on the fp64 MFMA; then the fp64-only engine for the narrow tiles; then engine i8-exact for every n.
mkdir -p gpurun_out/r03
{
echo "== default engine"
for n in 3 4 5 6 7 8 10 12 16 20 32 33 64 128; do python tools/perf_gemm.py 500000 50000 $n 10 2>&1 | grep tile; done
python tools/perf_gemm.py 1000000 50000 32 5 2>&1 | grep tile
echo "== MXA_ENGINE=f64-strict (fp64 arithmetic only: single-group MFMA tile for n <= 4, no peel)"
for n in 3 4 5 6 10; do MXA_ENGINE=f64-strict python tools/perf_gemm.py 500000 50000 $n 10 2>&1 | grep tile; done
echo "== MXA_ENGINE=i8-exact"
for n in 8 10 12 16 20 32 33 64; do MXA_ENGINE=i8-exact python tools/perf_gemm.py 500000 50000 $n 10 2>&1 | grep tile; done
MXA_ENGINE=i8-exact python tools/perf_gemm.py 1000000 50000 32 5 2>&1 | grep tile
} > gpurun_out/r03/gemm_by_n_warm.txt 2>&1
cat gpurun_out/r03/gemm_by_n_warm.txt
