#!/bin/bash
# fabric-side read traffic (FETCH_SIZE x 2 KiB) of the opt-in int8 engines at C2, per k_gemm_i8 launch
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out/r03"; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
for eng in i8 i8-exact; do
  rm -rf "$O/i8t"
  MXA_ENGINE=$eng timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$O/i8t" -- python3 "$R/tools/perf_gemm.py" 1000000 50000 32 2 > "$O/i8t_$eng.log" 2>&1
  python3 - <<PY
import csv, glob, collections
agg = collections.OrderedDict(); names = {}
for src in glob.glob("$O/i8t/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(src)):
        if "k_gemm_i8" in r["Kernel_Name"]:
            agg[r["Dispatch_Id"]] = agg.get(r["Dispatch_Id"], 0.0) + float(r["Counter_Value"]); names[r["Dispatch_Id"]] = r["Kernel_Name"][:40]
v = [(names[k], round(x * 2048 / 1e9, 1)) for k, x in agg.items()]
print("$eng", v[-4:])
PY
  grep tile "$O/i8t_$eng.log" | cut -c1-110
  rm -rf "$O/i8t"
done
