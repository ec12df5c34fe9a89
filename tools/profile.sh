#!/bin/bash
# tools/profile.sh <target> [args] -- the one entry point for measurements on the GPU box:
#   gpurun --timeout 900 -- 'bash tools/profile.sh <target>'
# Everything is written under gpurun_out/<round>/ (scratch); what is worth keeping is copied to profiles/ by hand.
# Targets:
#   tests            all -m gpu tests + smoke()
#   bench            default bench.py line -> bench_n1.json (+ bench_detail.json)
#   stats            bench.py (no extra legs) under rocprofv3 --kernel-trace --stats -> kernel stats csv
#   pmc-mfma         SQ counter pass of k_gemm (MFMA busy, clock) via tools/pmc_mfma_util.py
#   harness [big]    the reference's benchmark.f90 (GPU mode, unmodified) with the phase clock (PRINT_LEVEL=1)
#   power            power + clock trace of k_gemm, k_crossprod_gang FP4 / int8 (tools/power_trace.py)
#   gram             CG step (config-5 shard): step time, kernel timeline
#   xprod            crossproduct at config 3, both engines
#   rehearse         the driver's N > 1 bench commands on one GPU (in-process virtual shards, launcher ranks over gloo)
set -o pipefail
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
R=${MXA_ROUND:-r06}
O=$PWD/gpurun_out/$R; mkdir -p "$O"
export TMPDIR=/tmp
T=$1; shift
case "$T" in
tests)
  timeout -k 10 1100 python -m pytest tests -x -q -m gpu "$@" > "$O/gputests_full.txt" 2>&1; rc=$?
  tail -c 6000 "$O/gputests_full.txt"; tail -6 "$O/gputests_full.txt" > "$O/gputests.txt"; [ $rc = 0 ] &&
  timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee -a "$O/gputests.txt" ;;
bench)
  timeout -k 10 1100 python bench.py "$@" > "$O/bench_n1.json" 2> "$O/bench_n1.stderr" ; rc=$?
  cp -f bench_detail.json "$O/bench_detail.json" 2>/dev/null
  tail -c 4500 "$O/bench_n1.json"; tail -3 "$O/bench_n1.stderr"; exit $rc ;;
stats)
  cd /tmp && timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/stats" -- python3 "$OLDPWD/bench.py" --no-pmc --no-configs --no-abi --no-alt-engine --no-cpu-baseline --steps 10 > "$O/stats_bench.json" 2> "$O/stats.stderr" &&
  f=$(find "$O/stats" -name '*kernel_stats.csv' | head -1) && cp "$f" "$O/bench_n1_kernel_stats.csv" && head -8 "$f" ;;
pmc-mfma)
  bash tools/gpu_pmc_mfma_util.sh "$@" ;;
harness)
  D=/tmp/refdata; mkdir -p $D
  if [ "$1" = big ]; then S=250000; I=50000; else S=100000; I=20000; fi
  python3 tools/make_bed_dataset.py $D/d $S $I || exit 1
  for i in 1 2; do
    ( cd $D && PRINT_LEVEL=1 OMP_NUM_THREADS=16 timeout -k 10 500 "$OLDPWD/oracle/_ref/fortran/benchmark.out" GPU d.bed d.freq > "$O/harness_${S}x${I}_run$i.txt" 2>&1 ) || exit 1
    grep -E "Elapsed time|Average time|dgemm_compressed '" "$O/harness_${S}x${I}_run$i.txt" | head -60
  done
  rm -rf $D ;;
power)
  [ -x tools/mfma_power_probe ] || hipcc -O3 --offload-arch=gfx950 -o tools/mfma_power_probe tools/mfma_power_probe.hip || exit 1
  for t in ${PT_TARGETS:-bare_f4 bare_i8 bare_f64 gemm xprod_f4 xprod_i8}; do
    timeout -k 10 300 python3 tools/power_trace.py $t ${1:-12} > "$O/power_trace_$t.txt" 2>&1 || { tail -5 "$O/power_trace_$t.txt"; exit 1; }
    tail -1 "$O/power_trace_$t.txt"
  done ;;
gram)
  timeout -k 10 600 python3 tools/perf_gram.py "$@" 2>&1 | tee "$O/gram.txt" ;;
xprod)
  for e in f4 i8; do
    if [ $e = i8 ]; then export MXA_XPROD_ENGINE=i8; else unset MXA_XPROD_ENGINE; fi
    timeout -k 10 500 python3 tools/perf_crossprod.py ${1:-500000} ${2:-100000} 2 2>&1 | tee -a "$O/xprod.txt" || exit 1
  done ;;
rehearse)
  bash tools/gpu_r5_rehearse_n.sh ;;
*)
  sed -n 2,18p "$0"; exit 2 ;;
esac
