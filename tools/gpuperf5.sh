cd $GRAFT_REPO_ROOT
for m in 0 1; do MXA_GEMM_MODE=$m timeout -k 10 300 python tools/perf_gemm.py 200000 50000 32 3 2>&1 | grep mode=; done
MXA_GEMM_MODE=1 timeout -k 10 500 python -m pytest tests/test_dgemm_gpu.py -x -q -m gpu 2>&1 | tail -3
