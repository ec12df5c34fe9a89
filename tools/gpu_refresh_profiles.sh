cd $GRAFT_REPO_ROOT
timeout -k 10 900 python bench.py --steps 5 --warmup 1 2>&1 | grep '^{' > gpurun_out/bench_unprofiled.json
cat gpurun_out/bench_unprofiled.json | cut -c1-330
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/prof_cur && mkdir -p $R/gpurun_out/prof_cur
timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_cur -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof_cur/bench.log 2>&1
grep '^{' $R/gpurun_out/prof_cur/bench.log | cut -c1-200
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $R/gpurun_out/pmc_$c; mkdir -p $R/gpurun_out/pmc_$c
  timeout -k 10 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$c -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-alt-engine > $R/gpurun_out/pmc_$c/run.log 2>&1
done
echo pmc done
