#!/bin/bash
# tools/profile.sh <target> [args] -- the one entry point for measurements on the GPU box:
#   gpurun --timeout 900 -- 'bash tools/profile.sh <target>'
# Everything is written under gpurun_out/<round>/ (scratch; MXA_ROUND, default r06); what is worth keeping is copied to profiles/ by hand.
# Targets:
#   tests [pytest args]  all -m gpu tests + smoke(); full log in gputests_full.txt
#   bench [bench args]   default bench.py line -> bench_n1.json (+ bench_detail.json)
#   stats                bench.py (headline only) under rocprofv3 --kernel-trace --stats -> bench_n1_kernel_stats.csv
#   pmc-mfma <gemm|i8|xprod_f4|xprod_i8>        SQ counter pass (MFMA busy, clock, waits) of one kernel family -> pmc_mfma_util_<t>.json
#   pmc-traffic <gemm|i8|gram|xprod_f4|xprod_i8>  FETCH_SIZE / WRITE_SIZE passes (separate runs) -> pmc_traffic_<t>.txt
#   harness [big]        the reference's benchmark.f90 (GPU mode, unmodified) with the phase clock of the plain ABI (PRINT_LEVEL=1)
#   power [seconds]      power + clock trace: bare MFMA loops (f4, i8, f64), k_gemm, k_crossprod_gang FP4 / int8 (tools/power_trace.py); PT_TARGETS="i8t1 i8tn1 gram hbm_plain hbm_tn": the
#                        int8 kernels of a CG step, the whole step, and their data movement alone
#   gram                 CG step (config-5 shard): step time + kernel timeline of three steps
#   xprod [snps indiv]   crossproduct kernel time at config 3, both engines
#   gemm <snps indiv n reps>   k_gemm / k_gemm_i8 kernel time of one shape, 'N' and 'T'
#   rehearse             the driver's N > 1 bench commands on one GPU (8 virtual shards in-process; 2 and 4 launcher ranks over gloo)
#   soak                 tools/soak.py + fuzz_shapes.py + fuzz_crossprod.py
set -o pipefail
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
R=$PWD
RND=${MXA_ROUND:-r06}
O=$R/gpurun_out/$RND; mkdir -p "$O"
export TMPDIR=/tmp
SQ="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS GRBM_GUI_ACTIVE"
T=$1; shift

# the program (after `--` of rocprofv3: python3 itself, never a wrapper) and the kernel-name filter of a kernel family
family() {
  case "$1" in
    gemm)     PROG="python3 $R/tools/perf_gemm.py 1000000 50000 32 2"; KERN="k_gemm<" ;;
    i8)       PROG="python3 $R/tools/perf_gemm.py 250000 100000 1 5"; KERN="k_gemm_i8" ;;
    gram)     PROG="python3 $R/tools/perf_gram.py 250000 100000 1"; KERN="mxa::" ;;
    xprod_f4) unset MXA_XPROD_ENGINE; PROG="python3 $R/tools/perf_crossprod.py 500000 100000 2"; KERN="k_crossprod_gang" ;;
    xprod_i8) export MXA_XPROD_ENGINE=i8; PROG="python3 $R/tools/perf_crossprod.py 500000 100000 2"; KERN="k_crossprod_gang" ;;
    *) echo "unknown kernel family $1"; exit 2 ;;
  esac
}

case "$T" in
tests)
  timeout -k 10 1100 python -m pytest tests -x -q -m gpu "$@" > "$O/gputests_full.txt" 2>&1; rc=$?
  tail -c 6000 "$O/gputests_full.txt"; tail -6 "$O/gputests_full.txt" > "$O/gputests.txt"; [ $rc = 0 ] &&
  timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee -a "$O/gputests.txt" ;;
bench)
  timeout -k 10 1100 python bench.py "$@" > "$O/bench_n1.json" 2> "$O/bench_n1.stderr"; rc=$?
  cp -f bench_detail.json "$O/bench_detail.json" 2>/dev/null
  tail -c 4500 "$O/bench_n1.json"; tail -3 "$O/bench_n1.stderr"; exit $rc ;;
stats)
  rm -rf "$O/stats"
  ( cd /tmp && timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/stats" -- python3 "$R/bench.py" --no-pmc --no-configs --no-abi --no-alt-engine --no-cpu-baseline --steps 10 > "$O/stats_bench.json" 2> "$O/stats.stderr" ) || { tail -5 "$O/stats.stderr"; exit 1; }
  f=$(find "$O/stats" -name '*kernel_stats.csv' | head -1) && cp "$f" "$O/bench_n1_kernel_stats.csv" && head -8 "$f"; rm -rf "$O/stats" ;;
pmc-mfma)
  family "${1:-gemm}"; D="$O/pmc_$1"; rm -rf "$D"; mkdir -p "$D"
  ( cd /tmp && timeout -k 10 700 rocprofv3 --pmc $SQ --kernel-trace --output-format csv -d "$D" -- $PROG > "$D/run.log" 2>&1 ) || { tail -5 "$D/run.log"; exit 1; }
  tail -2 "$D/run.log"; python3 tools/pmc_mfma_util.py "$D" "$KERN" "$O/pmc_mfma_util_$1.json"; rm -rf "$D"/*/ ;;
pmc-traffic)
  family "${1:-gemm}"
  for c in FETCH_SIZE WRITE_SIZE; do
    D="$O/pmc_${1}_$c"; rm -rf "$D"; mkdir -p "$D"
    ( cd /tmp && timeout -k 10 700 rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$D" -- $PROG > "$D/run.log" 2>&1 ) || { tail -5 "$D/run.log"; exit 1; }
  done
  python3 tools/pmc_traffic.py "$O/pmc_${1}_FETCH_SIZE" "$O/pmc_${1}_WRITE_SIZE" "$KERN" "$O/pmc_traffic_$1.json" 2>&1 | tee "$O/pmc_traffic_$1.txt"; rm -rf "$O"/pmc_${1}_*/*/ ;;
harness)
  D=/tmp/refdata; mkdir -p $D
  if [ "$1" = big ]; then S=250000; I=50000; else S=100000; I=20000; fi
  python3 tools/make_bed_dataset.py $D/d $S $I || exit 1
  for i in 1 2; do
    ( cd $D && PRINT_LEVEL=1 OMP_NUM_THREADS=16 timeout -k 10 500 "$R/oracle/_ref/fortran/benchmark.out" GPU d.bed d.freq > "$O/harness_${S}x${I}_run$i.txt" 2>&1 ) || exit 1
    grep -E "Elapsed time|Average time|dgemm_compressed '" "$O/harness_${S}x${I}_run$i.txt" | cut -c1-330 | head -50
  done
  rm -rf $D ;;
power)
  [ -x tools/mfma_power_probe ] || hipcc -O3 --offload-arch=gfx950 -o tools/mfma_power_probe tools/mfma_power_probe.hip || exit 1
  for t in ${PT_TARGETS:-bare_f4 bare_i8 bare_f64 gemm xprod_f4 xprod_i8}; do
    timeout -k 10 300 python3 tools/power_trace.py $t ${1:-12} > "$O/power_trace_$t.txt" 2>&1 || { tail -5 "$O/power_trace_$t.txt"; exit 1; }
    tail -1 "$O/power_trace_$t.txt"
  done ;;
gram)
  timeout -k 10 600 python3 tools/perf_gram.py 250000 100000 1 2>&1 | tee "$O/gram.txt" || exit 1
  D="$O/gram_trace"; rm -rf "$D"
  ( cd /tmp && timeout -k 10 600 rocprofv3 --kernel-trace --output-format csv -d "$D" -- python3 "$R/tools/perf_gram.py" 250000 100000 1 > "$O/gram_trace_run.log" 2>&1 ) || exit 1
  python3 - "$(find "$D" -name '*kernel_trace.csv' | head -1)" <<'PY' | tee "$O/gram_step_kernel_timeline.txt"
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "mxa::" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
tail = rows[-21:]
t0, prev = int(tail[0]["Start_Timestamp"]), None
for r in tail:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{(s - t0) / 1e3:9.1f} us  +{((s - prev) / 1e3 if prev else 0.0):6.1f} gap  {(e - s) / 1e3:8.1f} us  {r['Kernel_Name'].split('(')[0][:60]}")
    prev = e
PY
  rm -rf "$D" ;;
xprod)
  for e in f4 i8; do
    if [ $e = i8 ]; then export MXA_XPROD_ENGINE=i8; else unset MXA_XPROD_ENGINE; fi
    timeout -k 10 500 python3 tools/perf_crossprod.py ${1:-500000} ${2:-100000} 2 2>&1 | tee -a "$O/xprod.txt" || exit 1
  done ;;
gemm)
  timeout -k 10 600 python3 tools/perf_gemm.py "$@" 2>&1 | tee -a "$O/gemm.txt" ;;
rehearse)
  timeout -k 10 500 python bench.py --gpus 8 --steps 5 --warmup 1 > "$O/bench_inprocess_8_virtual_shards.json" 2> "$O/inprocess8.err" || { tail -5 "$O/inprocess8.err"; exit 1; }
  cp -f bench_detail.json "$O/bench_detail_inprocess_8_virtual_shards.json"
  for n in 2 4; do
    MXA_BENCH_SINGLE_DEVICE=1 MXA_BENCH_BACKEND=gloo timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 2957$n \
      bench.py --gpus $n --steps 5 --warmup 1 > "$O/launcher_${n}ranks.log" 2>&1 || { tail -5 "$O/launcher_${n}ranks.log"; exit 1; }
    grep '^{' "$O/launcher_${n}ranks.log" > "$O/bench_launcher_${n}_ranks_one_gpu_gloo.json"
  done
  for f in "$O"/bench_inprocess_8_virtual_shards.json "$O"/bench_launcher_*_ranks_one_gpu_gloo.json; do echo "== $f"; cat "$f"; done ;;
soak)
  timeout -k 10 500 python3 tools/soak.py 2>&1 | tail -5 | tee "$O/soak.txt" &&
  timeout -k 10 400 python3 tools/fuzz_shapes.py 2>&1 | tail -3 | tee -a "$O/soak.txt" &&
  timeout -k 10 400 python3 tools/fuzz_crossprod.py 2>&1 | tail -3 | tee -a "$O/soak.txt" ;;
*)
  sed -n 2,20p "$0"; exit 2 ;;
esac
