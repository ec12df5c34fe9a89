#!/bin/bash
set -e
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 800 python -m pytest tests/test_cg_gpu.py tests/test_shard_gpu.py -x -q > gpurun_out/t1.log 2>&1 || { tail -40 gpurun_out/t1.log; exit 1; }
tail -3 gpurun_out/t1.log
timeout -k 10 300 python examples/grm_solve_cg.py --snps 250000 --indiv 100000 --max-iter 20 > gpurun_out/cg.log 2>&1 || { tail -20 gpurun_out/cg.log; exit 1; }
tail -5 gpurun_out/cg.log
