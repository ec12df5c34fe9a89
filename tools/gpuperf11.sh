cd $GRAFT_REPO_ROOT
timeout -k 10 600 python tools/perf_crossprod.py 500000 100000 1 2>&1 | grep -E "crossprod|rror"
