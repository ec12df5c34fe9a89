#!/bin/bash
# round 5: rocprofv3 summaries that go to profiles/ -- kernel stats of the bench command (headline alone, and with the config legs), MFMA-pipe
# utilisation of the shipped k_gemm instantiation, the n = 1 chain on the config-5 shard, the unprofiled bench line
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out/r05p"; rm -rf "$O"; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
stats() {  # name, command...
  local name=$1; shift
  rm -rf "$O/tmp_$name"
  timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/tmp_$name" -- "$@" > "$O/${name}_run.log" 2>&1
  local f=$(find "$O/tmp_$name" -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp "$f" "$O/${name}_kernel_stats.csv"
  rm -rf "$O/tmp_$name"
}
stats bench_n1 python3 "$R/bench.py" --no-pmc --no-abi --no-configs --steps 10 --warmup 2
grep '^{' "$O/bench_n1_run.log" > "$O/bench_n1_profiled.json"
stats bench_n1_with_config_legs python3 "$R/bench.py" --no-pmc --no-abi --steps 5 --warmup 1
grep '^{' "$O/bench_n1_with_config_legs_run.log" > "$O/bench_n1_with_config_legs_profiled.json"
PMC="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS GRBM_GUI_ACTIVE"
timeout -k 10 600 rocprofv3 --pmc $PMC --kernel-trace --output-format csv -d "$O/pmc1" -- python3 "$R/bench.py" --steps 1 --warmup 1 --no-cpu-baseline --no-alt-engine --no-pmc --no-abi --no-configs > "$O/pmc1_run.log" 2>&1
cd "$R"
python3 tools/pmc_mfma_util.py "$O/pmc1" "k_gemm<8, 8, 3" "$O/pmc_mfma_util.json" > "$O/pmc_util.txt" 2>&1
rm -rf "$O/pmc1"
{ for n in 1 2; do CENTERED=1 python3 tools/perf_gemm.py 250000 100000 $n 20 2>&1 | grep tile; done; python3 tools/perf_gram.py 250000 100000 1 2>&1 | grep "G\*v"; } > "$O/small_n_config5_shard.txt"
python3 bench.py --steps 20 --warmup 5 > "$O/bench_n1.json" 2> "$O/bench_n1.err"
# kernel timeline of the CG step, default object (one copy) and MXA_SINGLE_ORIENTATION=0 (two copies)
for so in 1 0; do
  cd /tmp; rm -rf "$O/tmp_g"
  MXA_SINGLE_ORIENTATION=$so timeout -k 10 600 rocprofv3 --kernel-trace --output-format csv -d "$O/tmp_g" -- python3 "$R/tools/perf_gram.py" 250000 100000 1 > "$O/gram_trace_run.log" 2>&1
  kt=$(find "$O/tmp_g" -name "*kernel_trace.csv" | head -1)
  echo "== kernel timeline of mxa_gram_matvec (config-5 shard 250k x 100k, n = 1, centred), MXA_SINGLE_ORIENTATION=$so: three steps" >> "$O/gram_step_kernel_timeline.txt"
  python3 - "$kt" >> "$O/gram_step_kernel_timeline.txt" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "mxa::" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
per = 7 if "k_gemm_i8_tn" in " ".join(r["Kernel_Name"] for r in rows[-20:]) else 7
tail = rows[-3 * per:]
t0 = int(tail[0]["Start_Timestamp"]); prev = None
for r in tail:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev) / 1e3 if prev else 0.0
    print(f"{(s - t0) / 1e3:9.1f} us  +{gap:6.1f} gap  {(e - s) / 1e3:8.1f} us  {r['Kernel_Name'].split('(')[0][:60]}")
    prev = e
PY
  rm -rf "$O/tmp_g"; cd "$R"
done
cat "$O/gram_step_kernel_timeline.txt"
ls -la "$O"; cat "$O/pmc_util.txt" "$O/small_n_config5_shard.txt"; head -8 "$O/bench_n1_kernel_stats.csv" | cut -c1-200; cut -c1-400 "$O/bench_n1.json"
