#!/bin/bash
# A/B of non-temporal result stores in the crossproduct epilogue: default build (on) against build/xst0 (plain stores); alternating, config 3
# build the variant first: make -C miraculix_amd/csrc OUT=../../build/xst0 CXXFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-result -Wno-inline-asm -I../../include -DMXA_XPROD_NT_STORE=0"
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for lib in "" build/xst0/libmiraculix_amd.so; do
  [ -n "$lib" ] && export MIRACULIX_AMD_LIBRARY=$GRAFT_REPO_ROOT/$lib || unset MIRACULIX_AMD_LIBRARY
  echo "== library: ${lib:-default (nt stores)}"
  timeout -k 10 200 python tools/perf_crossprod.py 500000 100000 3 2>&1 | grep crossprod
  MXA_XPROD_ENGINE=i8 timeout -k 10 200 python tools/perf_crossprod.py 500000 100000 2 2>&1 | grep crossprod
  timeout -k 10 200 python tools/perf_crossprod.py 128 100000 3 2>&1 | grep crossprod
done
done
