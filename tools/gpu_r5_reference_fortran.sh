#!/bin/bash
# the reference's own Fortran tests (unmodified, oracle/Makefile.ref_fortran) and its benchmark harness against this library on one MI355X
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05ft; O=$GRAFT_REPO_ROOT/gpurun_out/r05ft
timeout -k 10 600 python -m pytest tests/test_reference_fortran_gpu.py -q -p no:cacheprovider -rs > $O/tests.log 2>&1; rc=$?
tail -8 $O/tests.log
[ $rc = 0 ] || exit $rc
D=/tmp/refdata; mkdir -p $D
python3 tools/make_bed_dataset.py $D/small 5003 1201 &&
( cd $D && OMP_NUM_THREADS=4 timeout -k 10 300 $GRAFT_REPO_ROOT/oracle/_ref/fortran/test_5codesapi.out small.bed small.freq > $O/test_5codesapi_n.txt 2>&1 ) &&
( cd $D && OMP_NUM_THREADS=4 timeout -k 10 300 $GRAFT_REPO_ROOT/oracle/_ref/fortran/test_5codesapi_t.out small.bed small.freq > $O/test_5codesapi_t.txt 2>&1 ) &&
( cd oracle/_ref/fortran && timeout -k 10 120 ./test_solve.out > $O/test_solve.txt 2>&1 ) || exit 1
grep -E "No MC error|Average time|Elapsed time - C|plink2compressed|using device" $O/test_5codesapi_n.txt | head -12
# the benchmark harness, GPU mode: 1 warm-up + 10 repetitions of 'n' and of 't', ncol = 10, centred, host B / C
( time python3 tools/make_bed_dataset.py $D/big 100000 20000 ) 2>&1 | grep real
( cd $D && time OMP_NUM_THREADS=16 timeout -k 10 900 $GRAFT_REPO_ROOT/oracle/_ref/fortran/benchmark.out GPU big.bed big.freq > $O/benchmark_gpu_100k_x_20k.txt 2>&1 ) 2>&1 | grep real
grep -E "Average time|plink2compressed|transposition|using device" $O/benchmark_gpu_100k_x_20k.txt
rm -rf $D
