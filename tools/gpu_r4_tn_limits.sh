#!/bin/bash
# round 4: what bounds k_gemm_i8_tn?  build/exp_tn_nocompute (-DMXA_I8_TN_EXP_NOCOMPUTE: DMA ring + barriers + epilogue only, wrong results) against the shipped kernel
cd $GRAFT_REPO_ROOT
export MXA_I8_TN=1
for lib in "" build/exp_tn_nocompute/libmiraculix_amd.so "" build/exp_tn_nocompute/libmiraculix_amd.so; do
  [ -n "$lib" ] && export MIRACULIX_AMD_LIBRARY=$GRAFT_REPO_ROOT/$lib || unset MIRACULIX_AMD_LIBRARY
  echo "== library: ${lib:-default}"
  CENTERED=1 timeout -k 10 120 python tools/perf_gemm.py 250000 100000 1 20 2>&1 | grep tile
done
