cd $GRAFT_REPO_ROOT
CENTERED=1 timeout -k 10 600 python tools/perf_gemm.py 625000 200000 128 2 2>&1 | grep -E "mode=|rror"
CENTERED=1 timeout -k 10 600 python tools/perf_gemm.py 250000 100000 1 5 2>&1 | grep -E "mode=|rror"
CENTERED=1 timeout -k 10 600 python tools/perf_gemm.py 250000 100000 10 5 2>&1 | grep -E "mode=|rror"
