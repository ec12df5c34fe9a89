#!/bin/bash
# examples/fortran/gblup_cg.f90 (Fortran on the additive entry points) on one MI355X: a small data set with the dense check, and 250k x 50k
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05ft; O=$GRAFT_REPO_ROOT/gpurun_out/r05ft; D=/tmp/refdata; mkdir -p $D
python3 tools/make_bed_dataset.py $D/small 3001 803 && ( cd $D && timeout -k 10 120 $GRAFT_REPO_ROOT/examples/fortran/gblup_cg.out small.bed 1.0 200 ) > $O/fortran_cg_small.txt 2>&1 || { cat $O/fortran_cg_small.txt; exit 1; }
python3 tools/make_bed_dataset.py $D/big 250000 50000 && ( cd $D && timeout -k 10 300 $GRAFT_REPO_ROOT/examples/fortran/gblup_cg.out big.bed 1.0 200 ) > $O/fortran_cg_250k_x_50k.txt 2>&1 || { cat $O/fortran_cg_250k_x_50k.txt; exit 1; }
cat $O/fortran_cg_small.txt $O/fortran_cg_250k_x_50k.txt; rm -rf $D
