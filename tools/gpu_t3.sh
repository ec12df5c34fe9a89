#!/bin/bash
set -e
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 300 python tools/perf_gram.py 250000 100000 1 2>&1 | tail -2
timeout -k 10 300 python examples/grm_solve_cg.py --snps 250000 --indiv 100000 --max-iter 20 2>&1 | tail -1
