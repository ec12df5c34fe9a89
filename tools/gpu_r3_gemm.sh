#!/bin/bash
# round 3: k_gemm at C2 (plain + MXA_DIAG) and by n on 500k x 50k
mkdir -p gpurun_out/r03
{
python tools/perf_gemm.py 1000000 50000 32 5 2>&1 | grep tile
MXA_GEMM_TAIL_US=0 python tools/perf_gemm.py 1000000 50000 32 5 2>&1 | grep tile | sed 's/^/tail_us=0: /'
MXA_DIAG=1 python tools/perf_gemm.py 1000000 50000 32 1 2>&1 | grep -E "MXA_DIAG" | tail -8
for n in 3 4 5 8 10 12 16 20 33 64 128; do python tools/perf_gemm.py 500000 50000 $n 5 2>&1 | grep tile; done
} > gpurun_out/r03/gemm_perf.txt 2>&1
cat gpurun_out/r03/gemm_perf.txt
