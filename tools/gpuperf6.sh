cd $GRAFT_REPO_ROOT
for m in 0 1; do MXA_DIAG=1 MXA_GEMM_MODE=$m timeout -k 10 300 python tools/perf_gemm.py 200000 50000 32 1 2>&1 | grep -E "mode=|MXA_DIAG" | tail -4; done
