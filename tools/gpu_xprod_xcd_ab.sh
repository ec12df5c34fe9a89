#!/bin/bash
# A/B of the XCD-aware tile order of k_crossprod2: kernel time and FETCH_SIZE at 500k x 49152
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R && timeout -k 10 300 python -m pytest tests/test_crossprod_gpu.py -x -q 2>&1 | tail -2
cd /tmp
for x in 1 0; do
  MXA_XPROD_XCD=$x timeout -k 10 300 python3 $R/tools/perf_crossprod.py 500000 49152 2 2>&1 | grep -v amdgpu | sed "s/^/xcd_tiles=$x /"
  rm -rf $R/gpurun_out/xx_$x; mkdir -p $R/gpurun_out/xx_$x
  MXA_XPROD_XCD=$x timeout -k 10 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/xx_$x -- python3 $R/tools/perf_crossprod.py 500000 49152 1 > $R/gpurun_out/xx_$x/run.log 2>&1
  python3 - <<PY
import csv, glob, collections
agg = collections.OrderedDict()
for r in csv.DictReader(open(glob.glob("$R/gpurun_out/xx_$x/**/*_counter_collection.csv", recursive=True)[0])):
    if "k_crossprod2" in r["Kernel_Name"]:
        agg[r["Dispatch_Id"]] = agg.get(r["Dispatch_Id"], 0.0) + float(r["Counter_Value"])
print("xcd_tiles=$x FETCH_SIZE raw GB per k_crossprod2 launch:", [round(v * 1024 / 1e9, 1) for v in agg.values()])
PY
done
