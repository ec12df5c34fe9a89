cd $GRAFT_REPO_ROOT
timeout -k 10 500 python -m pytest tests/test_crossprod_gpu.py -x -q -m gpu 2>&1 | tail -2
MXA_DIAG=1 timeout -k 10 300 python tools/perf_crossprod.py 131072 32768 1 2>&1 | grep -E "MXA_DIAG|rror" | tail -1
timeout -k 10 300 python tools/perf_crossprod.py 131072 32768 2 2>&1 | grep -E "crossprod|rror"
timeout -k 10 300 python tools/perf_crossprod.py 500000 49152 1 2>&1 | grep -E "crossprod|rror"
