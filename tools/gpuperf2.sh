cd $GRAFT_REPO_ROOT
for m in 0 3 4 5; do MXA_GEMM_MODE=$m timeout -k 10 300 python tools/perf_gemm.py 200000 50000 32 3 2>&1 | grep mode=; done
