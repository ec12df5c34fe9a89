#!/bin/bash
# round 4: k_gemm_i8_tn with 256-individual strips (MXA_I8_TN_SLABS=2: 4 waves, two workgroups per CU) against 512 (=4: 8 waves, one per CU) and the plain kernel, config-5 shard
cd $GRAFT_REPO_ROOT
for v in "0 2" "1 4" "1 2" "1 4" "1 2"; do
  set -- $v; export MXA_I8_TN=$1 MXA_I8_TN_SLABS=$2
  echo "== MXA_I8_TN=$1 MXA_I8_TN_SLABS=$2"
  for n in 1 2; do CENTERED=1 timeout -k 10 120 python tools/perf_gemm.py 250000 100000 $n 20 2>&1 | grep tile; done
  timeout -k 10 120 python tools/perf_gram.py 250000 100000 1 2>&1 | grep "gram"
done
unset MXA_I8_TN
for sl in 4 2; do echo "== single-orientation object, MXA_I8_TN_SLABS=$sl"; MXA_I8_TN_SLABS=$sl MXA_SINGLE_ORIENTATION=1 timeout -k 10 120 python tools/perf_gram.py 250000 100000 1 2>&1 | grep "G\*v"; done
