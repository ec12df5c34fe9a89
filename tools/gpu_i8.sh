#!/bin/bash
# opt-in int8 engine: parity tests, kernel timing over n, in-kernel clocks (MXA_DIAG)
set -e
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_engine_i8_gpu.py -x -q > gpurun_out/i8_test.log 2>&1 || { tail -30 gpurun_out/i8_test.log; exit 1; }
tail -3 gpurun_out/i8_test.log
: > gpurun_out/i8_perf.log
for n in 32 16 10 4 1 64; do
MXA_ENGINE=i8 timeout -k 10 300 python tools/perf_gemm.py 1000000 50000 $n 5 2>&1 | grep -v amdgpu.ids >> gpurun_out/i8_perf.log
done
MXA_DIAG=1 MXA_ENGINE=i8 timeout -k 10 300 python tools/perf_gemm.py 1000000 50000 32 1 2>&1 | grep DIAG | sort -u | head -4 >> gpurun_out/i8_perf.log
cat gpurun_out/i8_perf.log
