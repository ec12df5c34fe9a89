#!/bin/bash
# config 3 crossproduct: the gang-synchronised kernel with 0 / 1 / 3 meetings inside a tile (MXA_XPROD_GANG_MID = parts), alternating; then the fabric-side read traffic
for v in 1 2 4 1 2 4; do
  echo "MXA_XPROD_GANG_MID=$v"; MXA_XPROD_GANG_MID=$v python tools/perf_crossprod.py 500000 100000 3 2>&1 | grep crossprod
done
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out/r04x"; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
for v in 1 2 4; do
  rm -rf "$O/xt_$v"
  MXA_XPROD_GANG_MID=$v timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$O/xt_$v" -- python3 "$R/tools/perf_crossprod.py" 500000 100000 1 > "$O/xt_$v.log" 2>&1
  python3 - <<PY
import csv, glob, collections
agg = collections.OrderedDict()
for src in glob.glob("$O/xt_$v/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(src)):
        if "k_crossprod" in r["Kernel_Name"]:
            agg[r["Dispatch_Id"]] = agg.get(r["Dispatch_Id"], 0.0) + float(r["Counter_Value"])
v = list(agg.values())
print("MXA_XPROD_GANG_MID=$v FETCH_SIZE: %d dispatches, %s GB per launch (x2 correction applied)" % (len(v), ", ".join("%.1f" % (x * 2048.0 / 1e9) for x in v)))
PY
  rm -rf "$O/xt_$v"
done
