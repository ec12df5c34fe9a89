#!/bin/bash
# fabric-side read traffic of k_gemm_i8 on the config-5 shard (n = 1, 2, 4): algorithmic 6.25 GB of packed matrix + the digit slabs
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out/r04s"; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
for n in 1 4; do
  rm -rf "$O/snt"
  CENTERED=1 timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$O/snt" -- python3 "$R/tools/perf_gemm.py" 250000 100000 $n 3 > "$O/snt_$n.log" 2>&1
  python3 - <<PY
import csv, glob, collections
agg = collections.OrderedDict(); names = {}
for src in glob.glob("$O/snt/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(src)):
        if "k_gemm_i8" in r["Kernel_Name"]:
            agg[r["Dispatch_Id"]] = agg.get(r["Dispatch_Id"], 0.0) + float(r["Counter_Value"]); names[r["Dispatch_Id"]] = r["Kernel_Name"][:36]
v = [(names[k], round(x * 2048 / 1e9, 2)) for k, x in agg.items()]
print("n=$n", v[-4:])
PY
  rm -rf "$O/snt"
done
