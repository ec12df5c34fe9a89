cd $GRAFT_REPO_ROOT
for m in 0 1 2; do MXA_GEMM_MODE=$m timeout -k 10 300 python tools/perf_gemm.py 400000 50000 32 4 2>&1 | grep mode=; done
