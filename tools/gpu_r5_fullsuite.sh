#!/bin/bash
# the whole GPU suite (no -x: list every failure), then smoke
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 1150 python -m pytest tests -q -m gpu -rf -p no:cacheprovider > gpurun_out/r5_fullsuite.log 2>&1
rc=$?
grep -E "^FAILED|^ERROR|passed|failed" gpurun_out/r5_fullsuite.log | tail -40
exit $rc
