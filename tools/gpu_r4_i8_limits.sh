#!/bin/bash
# round 4: what bounds k_gemm_i8<1,2,1> at n = 1 (config-5 shard: 0.955 ms for 6.26 GB = 6.55 TB/s against the 6.9-7.0 TB/s a bare non-temporal LDS-DMA stream reads)?
# variants (wrong results, timing only): build/exp_nomfma = operands formed, no MFMA; build/exp_cheapb = the digit slabs of stages 0/1 over and over (always L2 hits)
#   F="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-result -Wno-inline-asm -I../../include"
#   make -C miraculix_amd/csrc OUT=../../build/exp_nomfma CXXFLAGS="$F -DMXA_I8_EXP_NOMFMA"; make -C miraculix_amd/csrc OUT=../../build/exp_cheapb CXXFLAGS="$F -DMXA_I8_EXP_CHEAPB"
cd $GRAFT_REPO_ROOT
for lib in "" build/exp_nomfma/libmiraculix_amd.so build/exp_cheapb/libmiraculix_amd.so; do
  [ -n "$lib" ] && export MIRACULIX_AMD_LIBRARY=$GRAFT_REPO_ROOT/$lib || unset MIRACULIX_AMD_LIBRARY
  echo "== library: ${lib:-default}"
  for n in 1 2; do CENTERED=1 timeout -k 10 120 python tools/perf_gemm.py 250000 100000 $n 20 2>&1 | grep tile; done
done
unset MIRACULIX_AMD_LIBRARY
echo "== default library, splits of the 'N' product forced (MXA_I8_SPLITS)"
for sp in 2 3 4 5 8 10; do MXA_I8_SPLITS=$sp CENTERED=1 timeout -k 10 120 python tools/perf_gemm.py 250000 100000 1 20 2>&1 | grep "tile.* N "; done
echo "== bare stream"
[ -x tools/hbm_read_probe ] && timeout -k 10 120 tools/hbm_read_probe 2>&1 | tail -n 12
