cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R && MXA_BENCH_FORCE_DIST=1 timeout -k 10 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | tail -1
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $R/gpurun_out/pmc_$c; mkdir -p $R/gpurun_out/pmc_$c
  timeout -k 10 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$c -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $R/gpurun_out/pmc_$c/run.log 2>&1
  tail -1 $R/gpurun_out/pmc_$c/run.log | cut -c1-200
done
