#!/bin/bash
set -e
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 900 python bench.py > gpurun_out/bench.json 2> gpurun_out/bench.err || { tail -20 gpurun_out/bench.err; exit 1; }
cat gpurun_out/bench.json
