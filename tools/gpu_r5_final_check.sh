#!/bin/bash
# smoke() and the default bench line on the final code
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05f
timeout -k 10 300 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r05f/smoke.log 2>&1 && tail -2 gpurun_out/r05f/smoke.log &&
( time timeout -k 10 800 python3 bench.py > gpurun_out/r05f/bench_default.json 2> gpurun_out/r05f/bench_default.err ) 2>&1 | tail -3 &&
cut -c1-600 gpurun_out/r05f/bench_default.json
