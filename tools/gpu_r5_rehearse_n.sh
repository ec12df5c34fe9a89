#!/bin/bash
# round 5: the driver's N > 1 commands rehearsed on a ONE-GPU box with the final code (control flow and line format, not scaling numbers):
#   (a) python bench.py --gpus 8 (no launcher: 8 virtual shards behind the C ABI), (b) the launcher form with 2 and 4 ranks on cuda:0 over gloo
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out/r05n"; rm -rf "$O"; mkdir -p "$O"; cd "$R"
timeout -k 10 500 python bench.py --gpus 8 --steps 5 --warmup 1 > "$O/bench_inprocess_8_virtual_shards.json" 2> "$O/inprocess8.err"
for n in 2 4; do
  MXA_BENCH_SINGLE_DEVICE=1 MXA_BENCH_BACKEND=gloo timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 2957$n \
    bench.py --gpus $n --steps 5 --warmup 1 > "$O/launcher_${n}ranks_one_gpu_gloo.log" 2>&1
  grep '^{' "$O/launcher_${n}ranks_one_gpu_gloo.log" > "$O/bench_launcher_${n}_ranks_one_gpu_gloo.json"
done
ls -la "$O"; for f in "$O"/*.json; do echo "== $f"; cut -c1-700 "$f"; done; tail -n 3 "$O"/*.err
