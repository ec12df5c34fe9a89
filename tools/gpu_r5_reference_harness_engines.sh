#!/bin/bash
# the reference's benchmark harness (unmodified, GPU mode, ncol = 10, centred, host B / C) under the default engine and under MXA_ENGINE=i8-exact (opt-in, exact int8 slicing for every n)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05ft; O=$GRAFT_REPO_ROOT/gpurun_out/r05ft; D=/tmp/refdata; mkdir -p $D
python3 tools/make_bed_dataset.py $D/big 250000 50000 || exit 1
for e in f64 i8-exact; do
  ( cd $D && MXA_ENGINE=$e OMP_NUM_THREADS=4 timeout -k 10 600 $GRAFT_REPO_ROOT/oracle/_ref/fortran/benchmark.out GPU big.bed big.freq > $O/benchmark_gpu_250k_x_50k_engine_$e.txt 2>&1 ) || exit 1
  echo "== MXA_ENGINE=$e"; grep -E "Average time" $O/benchmark_gpu_250k_x_50k_engine_$e.txt
done
rm -rf $D
