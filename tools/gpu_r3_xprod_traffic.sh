#!/bin/bash
# HBM-side read / write traffic of the crossproduct kernels at config 3 (FETCH_SIZE x 2 x 1 KiB, WRITE_SIZE x 1 KiB; separate --pmc passes)
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out/r03"; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf "$O/xt_$c"
  timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$O/xt_$c" -- python3 "$R/tools/perf_crossprod.py" ${XK:-500000} 100000 1 > "$O/xt_$c.log" 2>&1
done
python3 - <<PY
import csv, glob, collections
for c, fac in (("FETCH_SIZE", 2048.0), ("WRITE_SIZE", 1024.0)):
    agg = collections.OrderedDict()
    for src in glob.glob("$O/xt_%s/**/*_counter_collection.csv" % c, recursive=True):
        for r in csv.DictReader(open(src)):
            if "k_crossprod" in r["Kernel_Name"]:
                agg[r["Dispatch_Id"]] = agg.get(r["Dispatch_Id"], 0.0) + float(r["Counter_Value"])
    v = list(agg.values())
    print(c, "%d dispatches, total %.1f GB" % (len(v), sum(v) * fac / 1e9), "(two product calls: halve)")
PY
rm -rf "$O/xt_FETCH_SIZE" "$O/xt_WRITE_SIZE"
