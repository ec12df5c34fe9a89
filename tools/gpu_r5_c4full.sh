#!/bin/bash
# round 5: the grouped-split / incremental-staging tests, then BASELINE config 4 at its full size as one product (tests/test_config4_full_gpu.py)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_grouped_and_incremental_gpu.py -x -q -m gpu > gpurun_out/r5_grouped.log 2>&1 || { tail -30 gpurun_out/r5_grouped.log; exit 1; }
tail -3 gpurun_out/r5_grouped.log
timeout -k 10 900 python -m pytest tests/test_config4_full_gpu.py -x -q -m gpu -s > gpurun_out/r5_c4full.log 2>&1
rc=$?
tail -40 gpurun_out/r5_c4full.log
exit $rc
