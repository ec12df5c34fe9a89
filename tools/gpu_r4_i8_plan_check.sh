#!/bin/bash
# round 4: the refitted split plan of k_gemm_i8 across shapes (plan's choice): n = 1..6, config-5 shard and 500k x 50k, both products
cd $GRAFT_REPO_ROOT
for n in 1 2; do CENTERED=1 timeout -k 10 120 python tools/perf_gemm.py 250000 100000 $n 20 2>&1 | grep tile; done
for n in 3 4 5 6; do timeout -k 10 120 python tools/perf_gemm.py 500000 50000 $n 10 2>&1 | grep tile; done
CENTERED=1 timeout -k 10 120 python tools/perf_gemm.py 2000000 100000 1 5 2>&1 | grep tile
for n in 1 4; do timeout -k 10 120 python tools/perf_gemm.py 60000 30000 $n 20 2>&1 | grep tile; done
timeout -k 10 120 python tools/perf_gram.py 250000 100000 1 2>&1 | grep "G\*v"
