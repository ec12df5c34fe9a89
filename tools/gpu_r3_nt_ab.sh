#!/bin/bash
# A/B of the non-temporal packed-matrix stream: default build (nt on for k_gemm_i8 / k_lut), build/nt0 (off), build/ntg (also on for k_gemm)
# build the variants first (in-tree, they travel with the snapshot): F="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-result -Wno-inline-asm -I../../include";
#   make -C miraculix_amd/csrc OUT=../../build/nt0 CXXFLAGS="$F -DMXA_NT_STREAM=0 -DMXA_NT_GEMM=0";  make -C miraculix_amd/csrc OUT=../../build/ntg CXXFLAGS="$F -DMXA_NT_GEMM=1"
cd $GRAFT_REPO_ROOT
for lib in "" build/nt0/libmiraculix_amd.so build/ntg/libmiraculix_amd.so; do
  [ -n "$lib" ] && export MIRACULIX_AMD_LIBRARY=$GRAFT_REPO_ROOT/$lib || unset MIRACULIX_AMD_LIBRARY
  echo "== library: ${lib:-default (nt stream on)}"
  for n in 1 2 4 6; do CENTERED=1 timeout -k 10 120 python tools/perf_gemm.py 250000 100000 $n 20 2>&1 | grep tile; done
  timeout -k 10 120 python tools/perf_gram.py 250000 100000 1 2>&1 | grep "G\*v"
  MXA_ENGINE=f64-strict CENTERED=1 timeout -k 10 120 python tools/perf_gemm.py 250000 100000 1 20 2>&1 | grep tile
  timeout -k 10 200 python tools/perf_gemm.py 1000000 50000 32 10 2>&1 | grep tile
done
