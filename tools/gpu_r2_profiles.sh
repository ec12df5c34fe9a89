#!/bin/bash
# round 2: rocprofv3 summaries that go to profiles/ (kernel stats of the bench command, crossproduct at config 3, n = 1 on the config-5 shard,
# MFMA-pipe utilisation of k_gemm and k_crossprod_f4)
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out/r02"; rm -rf "$O"; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
stats() {  # name, command...
  local name=$1; shift
  rm -rf "$O/tmp_$name"
  timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/tmp_$name" -- "$@" > "$O/${name}_run.log" 2>&1
  local f=$(find "$O/tmp_$name" -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp "$f" "$O/${name}_kernel_stats.csv"
  rm -rf "$O/tmp_$name"
}
stats bench_n1 python3 "$R/bench.py" --no-pmc --no-abi
grep '^{' "$O/bench_n1_run.log" > "$O/bench_n1_profiled.json"
stats crossprod_config3 python3 "$R/tools/perf_crossprod.py" 500000 100000 2
grep crossprod "$O/crossprod_config3_run.log" > "$O/crossprod_perf.txt"
MXA_XPROD_ENGINE=i8 python3 "$R/tools/perf_crossprod.py" 500000 100000 2 2>&1 | grep crossprod | sed 's/^/int8 engine: /' >> "$O/crossprod_perf.txt"
CENTERED=1 stats small_n1_config5_shard python3 "$R/tools/perf_gemm.py" 250000 100000 1 20
{ for n in 1 2; do CENTERED=1 python3 "$R/tools/perf_gemm.py" 250000 100000 $n 20 2>&1 | grep tile; done; python3 "$R/tools/perf_gram.py" 250000 100000 1 2>&1 | grep "G\*v"; \
  MXA_ENGINE=f64-strict CENTERED=1 python3 "$R/tools/perf_gemm.py" 250000 100000 1 20 2>&1 | grep tile | sed 's/^/f64-strict (k_lut): /'; } > "$O/small_n_config5_shard.txt"
{ for n in 4 5 8 10 12 16 20 33 128; do python3 "$R/tools/perf_gemm.py" 500000 50000 $n 5 2>&1 | grep tile; done; } > "$O/gemm_by_n_500k_x_50k.txt"
CENTERED=1 python3 "$R/tools/perf_gemm.py" 625000 200000 128 3 2>&1 | grep tile > "$O/gemm_config4_shard.txt"
python3 "$R/tools/perf_abi_host.py" 2>&1 | grep -E "dgemm|plink2" > "$O/abi_host_buffers.txt"
PMC="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS GRBM_GUI_ACTIVE"
rm -rf "$O/pmc1" "$O/pmc2"
timeout -k 10 600 rocprofv3 --pmc $PMC --kernel-trace --output-format csv -d "$O/pmc1" -- python3 "$R/bench.py" --steps 1 --warmup 1 --no-cpu-baseline --no-alt-engine --no-pmc --no-abi > "$O/pmc1_run.log" 2>&1
timeout -k 10 600 rocprofv3 --pmc $PMC --kernel-trace --output-format csv -d "$O/pmc2" -- python3 "$R/tools/perf_crossprod.py" 500000 49152 1 > "$O/pmc2_run.log" 2>&1
cd "$R"
python3 tools/pmc_mfma_util.py "$O/pmc1" "k_gemm<" "$O/pmc_mfma_util.json" > "$O/pmc_util.txt" 2>&1
python3 tools/pmc_mfma_util.py "$O/pmc2" "k_crossprod_f4" "$O/pmc_mfma_util_crossprod_f4.json" >> "$O/pmc_util.txt" 2>&1
rm -rf "$O/pmc1" "$O/pmc2"
python3 bench.py > "$O/bench_n1.json" 2> "$O/bench_n1.err"
ls -la "$O"; cat "$O/crossprod_perf.txt" "$O/small_n_config5_shard.txt" "$O/gemm_by_n_500k_x_50k.txt" "$O/gemm_config4_shard.txt" "$O/abi_host_buffers.txt" "$O/pmc_util.txt"; cut -c1-400 "$O/bench_n1.json"
