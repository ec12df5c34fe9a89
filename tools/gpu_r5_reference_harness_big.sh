#!/bin/bash
# the reference's benchmark harness (utils/benchmark/benchmark.f90, unmodified, GPU mode) against this library at a size where the GPU matters; twice at the small size (outlier position)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05ft; O=$GRAFT_REPO_ROOT/gpurun_out/r05ft; D=/tmp/refdata; mkdir -p $D
python3 tools/make_bed_dataset.py $D/mid 100000 20000 || exit 1
for i in 1 2; do ( cd $D && OMP_NUM_THREADS=16 timeout -k 10 600 $GRAFT_REPO_ROOT/oracle/_ref/fortran/benchmark.out GPU mid.bed mid.freq > $O/benchmark_gpu_100k_x_20k_run$i.txt 2>&1 ) || exit 1; grep -E "Elapsed time - Z - GPU" $O/benchmark_gpu_100k_x_20k_run$i.txt | tr '\n' ' '; echo; done
echo "generating 250k x 50k"; ( time python3 tools/make_bed_dataset.py $D/big 250000 50000 ) 2>&1 | grep real
( cd $D && time OMP_NUM_THREADS=16 timeout -k 10 1000 $GRAFT_REPO_ROOT/oracle/_ref/fortran/benchmark.out GPU big.bed big.freq > $O/benchmark_gpu_250k_x_50k.txt 2>&1 ) 2>&1 | grep real
grep -E "Elapsed time|Average time|plink2compressed|transposition|using device" $O/benchmark_gpu_250k_x_50k.txt
rm -rf $D
