#!/bin/bash
# round 3: fused GRM / LD epilogue -- parity tests, then timing against the unfused passes
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r03
timeout -k 10 600 python -m pytest tests/test_grm_ld_fused_gpu.py tests/test_crossprod_gpu.py tests/test_crossprod_engines_gpu.py -x -q -m gpu 2>&1 | tail -5 > gpurun_out/r03/grm_tests.txt
cat gpurun_out/r03/grm_tests.txt
grep -q passed gpurun_out/r03/grm_tests.txt && ! grep -q failed gpurun_out/r03/grm_tests.txt && timeout -k 10 500 python tools/perf_grm.py 500000 100000 40000 2>&1 | tee gpurun_out/r03/grm_perf.txt
