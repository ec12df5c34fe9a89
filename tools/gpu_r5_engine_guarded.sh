#!/bin/bash
# opt-in engine i8-guarded (mxa_set_engine(5)): by n on 500k x 50k against the default engine and i8-exact (one-copy objects, then two copies); the reference's Fortran
# harness under MXA_ENGINE=i8-guarded; the bench's opt-in legs
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05g; O=$GRAFT_REPO_ROOT/gpurun_out/r05g; : > $O/by_n.txt
for so in 1 0; do for n in 8 10 16 32; do for e in f64 i8-exact i8-guarded; do
  echo "== MXA_SINGLE_ORIENTATION=$so MXA_ENGINE=$e" >> $O/by_n.txt
  MXA_SINGLE_ORIENTATION=$so MXA_ENGINE=$e CENTERED=1 timeout -k 10 200 python3 tools/perf_gemm.py 500000 50000 $n 10 2>&1 | grep tile >> $O/by_n.txt || exit 1
done; done; done
cat $O/by_n.txt | cut -c1-150
D=/tmp/refdata; mkdir -p $D; python3 tools/make_bed_dataset.py $D/big 250000 50000 || exit 1
( cd $D && MXA_ENGINE=i8-guarded OMP_NUM_THREADS=4 timeout -k 10 600 $GRAFT_REPO_ROOT/oracle/_ref/fortran/benchmark.out GPU big.bed big.freq > $O/benchmark_gpu_250k_x_50k_engine_i8-guarded.txt 2>&1 ) || exit 1
grep -E "Elapsed time|Average time" $O/benchmark_gpu_250k_x_50k_engine_i8-guarded.txt | tr '\n' ' '; echo; rm -rf $D
timeout -k 10 600 python3 bench.py --no-pmc --no-abi --no-configs --no-cpu-baseline > $O/bench.json 2> $O/bench.err || exit 1
python3 - <<'PY'
import json
d = json.loads(open("gpurun_out/r05g/bench.json").read().strip().splitlines()[-1])
print("headline", d["value"], d["ms_per_step"])
for k in ("opt_in_engine", "opt_in_engine_exact", "opt_in_engine_guarded"):
    e = d[k]; print(k, e["value"], e["ms_per_step"], e["avg_kernel_ms"], e["max_colwise_rel_diff_vs_f64_engine"])
PY
