#!/bin/bash
# N = 2 control flow of bench.py on a one-GPU box: two ranks on cuda:0, gloo instead of RCCL (a rehearsal, not a measurement)
cd "$GRAFT_REPO_ROOT"
MXA_BENCH_SINGLE_DEVICE=1 MXA_BENCH_BACKEND=gloo timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29577 \
  bench.py --gpus 2 --steps 3 --warmup 1 --snps 400000 --indiv 30000 2>&1 | grep -E "^\{|rror|Traceback|assert" | cut -c1-900
