#!/bin/bash
# kernel averages of the CG step (config-5 shard) under rocprofv3
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05u; O=$GRAFT_REPO_ROOT/gpurun_out/r05u
cd /tmp && export TMPDIR=/tmp; rm -rf $O/tmp_g
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/tmp_g -- python3 $GRAFT_REPO_ROOT/tools/perf_gram.py 250000 100000 1 > $O/gram_run.log 2>&1
f=$(find $O/tmp_g -name "*kernel_stats.csv" | head -1)
python3 - $f <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "mxa::" in r["Name"]: print(r["Name"].split("(")[0][:50], r["Calls"], round(float(r["AverageNs"]) / 1e3, 2), "us")
PY
grep "G\*v" $O/gram_run.log; rm -rf $O/tmp_g
