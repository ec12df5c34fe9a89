cd $GRAFT_REPO_ROOT
for n in 1 2 4; do MXA_LUT_MAX_N=4 CENTERED=1 timeout -k 10 600 python tools/perf_gemm.py 250000 100000 $n 5 2>&1 | grep -E "mode=|rror"; done
