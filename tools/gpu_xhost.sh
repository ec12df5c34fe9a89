#!/bin/bash
set -e
cd "$GRAFT_REPO_ROOT"
timeout -k 10 600 python -m pytest tests/test_crossprod_gpu.py -x -q 2>&1 | tail -3
timeout -k 10 600 python tools/perf_crossprod_host.py 500000 60000 2>&1 | grep -v amdgpu
