#!/bin/bash
# A/B of the XCD-aware workgroup order of k_gemm: kernel time and FETCH_SIZE (HBM/fabric read traffic) per launch
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for x in 1 0; do
  MXA_XCD_ORDER=$x timeout -k 10 300 python3 $R/tools/perf_gemm.py 1000000 50000 32 5 2>&1 | grep -v amdgpu | sed "s/^/xcd_order=$x /"
  rm -rf $R/gpurun_out/xcd_$x; mkdir -p $R/gpurun_out/xcd_$x
  MXA_XCD_ORDER=$x timeout -k 10 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/xcd_$x -- python3 $R/tools/perf_gemm.py 1000000 50000 32 1 > $R/gpurun_out/xcd_$x/run.log 2>&1
  python3 - <<PY
import csv, glob, collections
agg = collections.OrderedDict()
for r in csv.DictReader(open(glob.glob("$R/gpurun_out/xcd_$x/**/*_counter_collection.csv", recursive=True)[0])):
    if "k_gemm<" in r["Kernel_Name"]:
        agg[r["Dispatch_Id"]] = agg.get(r["Dispatch_Id"], 0.0) + float(r["Counter_Value"])
print("xcd_order=$x FETCH_SIZE raw GB per k_gemm launch (N,N,T,T):", [round(v * 1024 / 1e9, 2) for v in agg.values()])
PY
done
