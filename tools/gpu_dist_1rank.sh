cd $GRAFT_REPO_ROOT
MXA_BENCH_FORCE_DIST=1 timeout -k 10 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/dist1.log 2>&1
grep -E "^\{|Error|error|Traceback" gpurun_out/dist1.log | cut -c1-300
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29544 bench.py --gpus 1 --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | grep -E "^\{|rror" | cut -c1-200
