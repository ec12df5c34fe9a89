#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_single_orientation_gpu.py tests/test_small_n_gpu.py -x -q -m gpu 2>&1 | tail -2
for n in 3 4 5 6 7; do python tools/perf_gemm.py 500000 50000 $n 10 2>&1 | grep "tile=.* N "; done
