#!/bin/bash
cd $GRAFT_REPO_ROOT
PRINT_LEVEL=1 timeout -k 10 400 python3 tools/perf_crossprod_calls.py 500000 100000 12 2>&1 | grep "crossproduct call\|== call" > gpurun_out/r05_crossprod_call_wall.txt
cat gpurun_out/r05_crossprod_call_wall.txt
