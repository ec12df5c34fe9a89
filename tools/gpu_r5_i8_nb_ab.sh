#!/bin/bash
# experiment: ring depth of the one-tile plain int8 kernel at EQUAL residency -- NB = 3 with the LDS padded to 48 KiB (three workgroups per CU) against NB = 4 (48 KiB);
# NB = 3 unpadded (four per CU, shipped) beside them
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out/r5_i8_half_mfma.txt; : > $O
for rep in 1 2; do
for shape in "250000 100000"; do
  for v in ship half; do
    lib=$GRAFT_REPO_ROOT/exp_build/lib_$v.so; [ $v = ship ] && lib=$GRAFT_REPO_ROOT/miraculix_amd/lib/libmiraculix_amd.so
    echo "== $v $shape" >> $O
    MIRACULIX_AMD_LIBRARY=$lib MXA_SINGLE_ORIENTATION=0 CENTERED=1 timeout -k 10 120 python3 tools/perf_gemm.py $shape 1 30 2>&1 | grep tile >> $O || exit 1
  done
done
done
cat $O
