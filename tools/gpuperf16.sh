cd $GRAFT_REPO_ROOT
timeout -k 10 500 python -m pytest tests/test_crossprod_gpu.py -x -q -m gpu 2>&1 | tail -2
for v in 1 2; do MXA_XPROD_VER=$v timeout -k 10 300 python tools/perf_crossprod.py 131072 32768 2 2>&1 | grep -E "crossprod|rror" | sed "s/^/v$v /"; done
MXA_XPROD_VER=2 timeout -k 10 300 python tools/perf_crossprod.py 500000 49152 1 2>&1 | grep -E "crossprod|rror"
