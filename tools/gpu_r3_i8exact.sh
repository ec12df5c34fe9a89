#!/bin/bash
# round 3: opt-in engine i8-exact (digit count chosen per call from the exponent span of B) by n on 500k x 50k and at C2, beside the default engine
mkdir -p gpurun_out/r03
{
echo "== engine i8-exact"
for n in 3 4 5 8 10 12 16 20 32 33 64; do MXA_ENGINE=i8-exact python tools/perf_gemm.py 500000 50000 $n 5 2>&1 | grep tile; done
MXA_ENGINE=i8-exact python tools/perf_gemm.py 1000000 50000 32 5 2>&1 | grep tile
echo "== default engine"
for n in 3 4 5 8 10 12 16 32; do python tools/perf_gemm.py 500000 50000 $n 5 2>&1 | grep tile; done
} > gpurun_out/r03/gemm_i8exact_perf.txt 2>&1
cat gpurun_out/r03/gemm_i8exact_perf.txt
