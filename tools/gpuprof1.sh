set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/prof1
timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof1 -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof1/bench.log 2>&1
find $R/gpurun_out/prof1 -name "*stats*" | head
for f in $(find $R/gpurun_out/prof1 -name "*kernel_stats.csv"); do cat $f; done
tail -2 $R/gpurun_out/prof1/bench.log
