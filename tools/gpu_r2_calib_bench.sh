#!/bin/bash
# round 2: FETCH_SIZE calibration on known-size streams, then the default bench.py run (which measures its own traffic)
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out"; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -o "TCC_EA0_RDREQ[A-Za-z0-9_]*" | sort -u > "$O/r2_rdreq_counters.txt"
rm -rf "$O/calib_fetch" "$O/calib_rdreq"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$O/calib_fetch" -- "$R/tools/pmc_calibrate" 8 > "$O/r2_calib_run.txt" 2>&1 || { tail -5 "$O/r2_calib_run.txt"; exit 1; }
timeout -k 10 300 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum --kernel-trace --output-format csv -d "$O/calib_rdreq" -- "$R/tools/pmc_calibrate" 8 >> "$O/r2_calib_run.txt" 2>&1 || echo "rdreq pass failed (counter names?)"
cd "$R" && CALIB_OUT="$O/r02_pmc_calibration.json" python tools/pmc_calibrate.py 8589934592 "$O/calib_fetch" "$O/calib_rdreq" > "$O/r2_calib_parse.txt" 2>&1
grep CALIB "$O/r2_calib_run.txt"; grep factor "$O/r02_pmc_calibration.json"
mkdir -p "$R/profiles" && cp "$O/r02_pmc_calibration.json" "$R/profiles/" 2>/dev/null
rm -rf "$O/calib_fetch" "$O/calib_rdreq"
cd "$R" && timeout -k 10 900 python bench.py > "$O/r2_bench.json" 2> "$O/r2_bench.err" || { tail -20 "$O/r2_bench.err"; exit 1; }
cat "$O/r2_bench.json"
