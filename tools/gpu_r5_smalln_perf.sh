#!/bin/bash
# round 5: the guarded small-n routes after the device-side verdict: CG step (config-5 shard) and n = 3..6 (500k x 50k), two copies against one copy; kernel timeline of the step
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
{
for so in 0 1; do
  echo "== MXA_SINGLE_ORIENTATION=$so: config-5 shard 250k x 100k, centred"
  MXA_SINGLE_ORIENTATION=$so CENTERED=1 timeout -k 10 200 python tools/perf_gemm.py 250000 100000 1 20 2>&1 | grep "tile="
  MXA_SINGLE_ORIENTATION=$so timeout -k 10 200 python tools/perf_gram.py 250000 100000 1 2>&1 | grep "G\*v"
  MXA_SINGLE_ORIENTATION=$so timeout -k 10 200 python tools/perf_gram.py 250000 100000 1 2>&1 | grep "G\*v"
  for n in 3 4 5 6 10; do
    MXA_SINGLE_ORIENTATION=$so timeout -k 10 200 python tools/perf_gemm.py 500000 50000 $n 10 2>&1 | grep "tile="
  done
done
} > gpurun_out/r5_smalln_perf.txt 2>&1
cat gpurun_out/r5_smalln_perf.txt
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out/r05g"; mkdir -p "$O"
for so in 0 1; do
cd /tmp && export TMPDIR=/tmp
rm -rf "$O/tmp_g"
MXA_SINGLE_ORIENTATION=$so timeout -k 10 600 rocprofv3 --kernel-trace --output-format csv -d "$O/tmp_g" -- python3 "$R/tools/perf_gram.py" 250000 100000 1 > "$O/gram_trace_run.log" 2>&1
kt=$(find "$O/tmp_g" -name "*kernel_trace.csv" | head -1)
echo "== kernel timeline of mxa_gram_matvec, MXA_SINGLE_ORIENTATION=$so" >> $R/gpurun_out/r5_gram_timeline.txt
python3 - "$kt" >> $R/gpurun_out/r5_gram_timeline.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows = [r for r in rows if "mxa::" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
tail = rows[-27:]
t0 = int(tail[0]["Start_Timestamp"])
prev_end = None
for r in tail:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev_end) / 1e3 if prev_end else 0.0
    print(f"{(s - t0) / 1e3:9.1f} us  +{gap:6.1f} gap  {(e - s) / 1e3:8.1f} us  {r['Kernel_Name'].split('(')[0][:60]}")
    prev_end = e
PY
rm -rf "$O/tmp_g"
done
cat $R/gpurun_out/r5_gram_timeline.txt
