#!/bin/bash
# kernel timeline of one dgemm_compressed call at small n (default 500k x 50k, n = 6): where the time outside the main kernel goes
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out/r04t"; mkdir -p "$O"
N=${1:-6}
cd /tmp && export TMPDIR=/tmp
rm -rf "$O/tmp_g"
timeout -k 10 600 rocprofv3 --kernel-trace --output-format csv -d "$O/tmp_g" -- python3 "$R/tools/perf_gemm.py" 500000 50000 $N 6 > "$O/trace_run.log" 2>&1
kt=$(find "$O/tmp_g" -name "*kernel_trace.csv" | head -1)
python3 - "$kt" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows = [r for r in rows if "mxa::" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
tail = rows[-14:]
t0 = int(tail[0]["Start_Timestamp"]); prev_end = None
for r in tail:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev_end) / 1e3 if prev_end else 0.0
    print(f"{(s - t0) / 1e3:9.1f} us  +{gap:6.1f} gap  {(e - s) / 1e3:8.1f} us  {r['Kernel_Name'].split('(')[0][:60]}")
    prev_end = e
PY
rm -rf "$O/tmp_g"
