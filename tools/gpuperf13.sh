cd $GRAFT_REPO_ROOT
timeout -k 10 500 python -m pytest tests/test_dgemm_gpu.py -x -q -m gpu 2>&1 | tail -3
for n in 1 2 4; do CENTERED=1 timeout -k 10 600 python tools/perf_gemm.py 250000 100000 $n 5 2>&1 | grep -E "mode=|rror"; done
