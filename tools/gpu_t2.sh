#!/bin/bash
set -e
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 800 python -m pytest tests/test_crossprod_gpu.py -x -q > gpurun_out/t2.log 2>&1 || { tail -40 gpurun_out/t2.log; exit 1; }
tail -3 gpurun_out/t2.log
timeout -k 10 300 python examples/grm_solve_cg.py --snps 250000 --indiv 100000 --max-iter 20 2>&1 | tail -2
timeout -k 10 300 python tools/perf_crossprod.py 500000 49152 2 2>&1 | tail -2
