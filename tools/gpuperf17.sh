cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_dgemm_gpu.py tests/test_staging_gpu.py tests/test_fullsize_gpu.py -x -q -m gpu 2>&1 | tail -3
for n in 1 2; do CENTERED=1 timeout -k 10 600 python tools/perf_gemm.py 250000 100000 $n 5 2>&1 | grep -E "mode=|rror"; done
timeout -k 10 600 python tools/perf_gemm.py 400000 50000 32 3 2>&1 | grep -E "mode=|rror"
