#!/bin/bash
set -e
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_sparse_gpu.py tests/test_abi_cpu.py tests/test_staging_gpu.py -x -q > gpurun_out/sparse_test.log 2>&1 || { tail -40 gpurun_out/sparse_test.log; exit 1; }
tail -3 gpurun_out/sparse_test.log
