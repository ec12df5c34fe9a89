#!/bin/bash
cd $GRAFT_REPO_ROOT
export MXA_I8_TN=1 MXA_I8_TN_SLABS=2
for sp in 2 3 4 5 6 7 9 13; do echo "MXA_I8_TN_SPLITS=$sp"; MXA_I8_TN_SPLITS=$sp CENTERED=1 timeout -k 10 120 python tools/perf_gemm.py 250000 100000 1 20 2>&1 | grep "tile.* N "; done
export MXA_I8_TN_SLABS=4
for sp in 5 9; do echo "SLABS=4 MXA_I8_TN_SPLITS=$sp"; MXA_I8_TN_SPLITS=$sp CENTERED=1 timeout -k 10 120 python tools/perf_gemm.py 250000 100000 1 20 2>&1 | grep "tile.* N "; done
