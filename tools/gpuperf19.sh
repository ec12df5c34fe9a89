cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_dgemm_gpu.py tests/test_edge_gpu.py tests/test_staging_gpu.py tests/test_shard_gpu.py -x -q -m gpu 2>&1 | tail -3
for n in 4 10 20 32 40; do timeout -k 10 600 python tools/perf_gemm.py 250000 100000 $n 3 2>&1 | grep -E "mode=|rror" | cut -c1-130; done
