#!/bin/bash
# opt-in int8 engines on default (one-copy) objects: 'N' at wide n in column chunks on k_gemm_i8_tn -- tests, then the bench's opt-in legs at C2
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05e
timeout -k 10 900 python -m pytest tests/test_engine_i8_gpu.py tests/test_engine_i8_exact_gpu.py tests/test_single_orientation_gpu.py -q -x -p no:cacheprovider > gpurun_out/r05e/tests.log 2>&1; rc=$?
tail -5 gpurun_out/r05e/tests.log
[ $rc = 0 ] || exit $rc
timeout -k 10 600 python3 bench.py --no-pmc --no-abi --no-configs --no-cpu-baseline > gpurun_out/r05e/bench.json 2> gpurun_out/r05e/bench.err || exit 1
python3 - <<'PY'
import json
d = json.loads(open("gpurun_out/r05e/bench.json").read().strip().splitlines()[-1])
print("headline", d["value"], d["ms_per_step"])
for k in ("opt_in_engine", "opt_in_engine_exact"):
    e = d[k]; print(k, e["value"], e["ms_per_step"], e["avg_kernel_ms"], e["digits_per_column"], e["int8_ops_per_s_P"], e["max_colwise_rel_diff_vs_f64_engine"])
PY
