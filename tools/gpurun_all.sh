# full GPU check: all gpu tests, smoke, bench + rocprof kernel stats
cd $GRAFT_REPO_ROOT
timeout -k 10 800 python -m pytest tests -x -q -m gpu 2>&1 | tail -5 || exit 1
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 || exit 1
timeout -k 10 900 python bench.py --steps 5 --warmup 1 2>&1 | tail -2 || exit 1
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/prof_cur && mkdir -p $R/gpurun_out/prof_cur
timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_cur -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof_cur/bench.log 2>&1
tail -1 $R/gpurun_out/prof_cur/bench.log
