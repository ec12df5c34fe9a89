cd $GRAFT_REPO_ROOT
timeout -k 10 300 python tools/perf_crossprod.py 131072 32768 2 2>&1 | grep crossprod
timeout -k 10 300 python tools/perf_crossprod.py 500000 49152 1 2>&1 | grep crossprod
