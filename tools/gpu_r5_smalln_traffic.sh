#!/bin/bash
# fabric-side read traffic of the HBM-bound int8 kernels on DEFAULT (one packed copy) objects: 'T' on k_gemm_i8, 'N' on k_gemm_i8_tn<4,1> (n = 1) and
# k_gemm_i8_tn<6,2> (n = 4, 6: two digit tiles per pass).  Algorithmic: one read of the packed matrix (6.26 GB with the tile padding) + the digit slabs.
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out/r05t"; mkdir -p "$O"; : > "$O/traffic.txt"
cd /tmp && export TMPDIR=/tmp
for cfg in "250000 100000 1" "500000 50000 4" "500000 50000 6"; do
  set -- $cfg
  rm -rf "$O/snt"
  CENTERED=1 timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$O/snt" -- python3 "$R/tools/perf_gemm.py" $1 $2 $3 3 > "$O/snt_$3.log" 2>&1 || exit 1
  python3 - >> "$O/traffic.txt" <<PY
import csv, glob, collections
agg = collections.OrderedDict(); names = {}
for src in glob.glob("$O/snt/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(src)):
        if "k_gemm_i8" in r["Kernel_Name"]:
            agg[r["Dispatch_Id"]] = agg.get(r["Dispatch_Id"], 0.0) + float(r["Counter_Value"]); names[r["Dispatch_Id"]] = r["Kernel_Name"].replace("void mxa::", "").split("(")[0][:40]
by = collections.OrderedDict()
for k, x in agg.items(): by.setdefault(names[k], []).append(round(x * 2048 / 1e9, 2))
for name, v in by.items(): print("$1 x $2 n=$3", name, "GB per launch, last three:", v[-3:])
PY
  rm -rf "$O/snt"
done
cat "$O/traffic.txt"
