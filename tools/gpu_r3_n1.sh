#!/bin/bash
# round 3: n = 1 on the config-5 shard: persistent int8 kernel parameter sweep
for wg in 4; do for piece in 100 160 250 400; do for tail in 0 16; do
  MXA_I8_WG_PER_CU=$wg MXA_I8_PIECE=$piece MXA_I8_TAIL=$tail CENTERED=1 python tools/perf_gemm.py 250000 100000 1 20 2>&1 | grep tile | sed "s/^/wg=$wg piece=$piece tail=$tail /" | cut -c1-170
done; done; done
