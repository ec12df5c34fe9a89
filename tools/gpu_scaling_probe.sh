cd $GRAFT_REPO_ROOT
for s in 500000 250000 125000; do timeout -k 10 300 python tools/perf_gemm.py $s 50000 32 10 2>&1 | grep -v amdgpu; done
MXA_BENCH_FORCE_DIST=1 timeout -k 10 600 python bench.py --snps 125000 --steps 10 --warmup 2 --no-cpu-baseline --no-alt-engine 2>&1 | grep '^{' | cut -c1-400
