cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_edge_gpu.py -x -q -m gpu 2>&1 | tail -8
timeout -k 10 600 python tools/perf_abi_host.py 2>&1 | grep -E "plink2compressed|dgemm_compressed|rror" > gpurun_out/abi_host.txt; cat gpurun_out/abi_host.txt
