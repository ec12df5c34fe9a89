#!/bin/bash
# full-size N = 1 bench line with all legs; prints the headline numbers and the wall time
mkdir -p gpurun_out/r03
T0=$(date +%s)
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r03/bench1.json 2> gpurun_out/r03/bench1.err
RC=$?
T1=$(date +%s)
echo "bench rc=$RC wall=$((T1-T0))s"
tail -3 gpurun_out/r03/bench1.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r03/bench1.json"))
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["traffic"])
for k in ("config5_cg_step", "config4_shard", "config3_crossprod", "abi_end_to_end", "cpu_baseline", "check"):
    print(k, json.dumps(d.get(k))[:1800])
PY
