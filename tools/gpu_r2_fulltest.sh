#!/bin/bash
# what the driver runs at round end (all -m gpu tests, smoke) + the rocprofv3 kernel stats of the bench command
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out/r02"; mkdir -p "$O"
cd "$R"
timeout -k 10 1000 python -m pytest tests -x -q -m gpu --durations=10 > "$O/gputest_all.log" 2>&1; tail -16 "$O/gputest_all.log"
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
cd /tmp && export TMPDIR=/tmp
rm -rf "$O/tmp_b"
timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/tmp_b" -- python3 "$R/bench.py" --no-pmc --no-abi > "$O/bench_n1_run.log" 2>&1
f=$(find "$O/tmp_b" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$O/bench_n1_kernel_stats.csv"; rm -rf "$O/tmp_b"
grep '^{' "$O/bench_n1_run.log" > "$O/bench_n1_profiled.json"; cut -c1-300 "$O/bench_n1_profiled.json"; head -2 "$O/bench_n1_kernel_stats.csv" | cut -c1-200
