cd $GRAFT_REPO_ROOT
for m in 2 4; do MXA_DIAG=1 MXA_GEMM_MODE=$m timeout -k 10 300 python tools/perf_gemm.py 200000 50000 32 1 2>&1 | grep -E "MXA_DIAG" | tail -1; MXA_GEMM_MODE=$m timeout -k 10 300 python tools/perf_gemm.py 200000 50000 32 3 2>&1 | grep -E "mode="; done
MXA_GEMM_MODE=4 timeout -k 10 500 python -m pytest tests/test_dgemm_gpu.py -x -q -m gpu 2>&1 | tail -3
