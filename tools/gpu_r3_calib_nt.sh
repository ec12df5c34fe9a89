#!/bin/bash
# round 3: FETCH_SIZE / RDREQ calibration including the non-temporal LDS-DMA stream (k_stream_dma<2>)
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out/r03"; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
rm -rf "$O/calib_fetch" "$O/calib_rdreq"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$O/calib_fetch" -- "$R/tools/pmc_calibrate" 8 > "$O/calib_run.txt" 2>&1 || { tail -5 "$O/calib_run.txt"; exit 1; }
timeout -k 10 300 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum --kernel-trace --output-format csv -d "$O/calib_rdreq" -- "$R/tools/pmc_calibrate" 8 >> "$O/calib_run.txt" 2>&1 || echo "rdreq pass failed"
cd "$R" && CALIB_OUT="$O/r03_pmc_calibration_nt.json" python tools/pmc_calibrate.py 8589934592 "$O/calib_fetch" "$O/calib_rdreq" > /dev/null 2>&1
rm -rf "$O/calib_fetch" "$O/calib_rdreq"
python3 - <<PY
import json
d = json.load(open("$O/r03_pmc_calibration_nt.json"))
for k, v in d.items():
    if isinstance(v, dict) and "FETCH_SIZE" in v:
        print(k, "FETCH_SIZE factor", round(v["FETCH_SIZE"]["bytes_over_counter_KiB"], 4), "request split / true", round(v.get("bytes_from_request_size_split_over_true", 0), 5))
PY
