cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/prof_x $R/gpurun_out/prof_l; mkdir -p $R/gpurun_out/prof_x $R/gpurun_out/prof_l
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_x -- python3 $R/tools/perf_crossprod.py 500000 100000 1 > $R/gpurun_out/prof_x/run.log 2>&1
grep crossprod $R/gpurun_out/prof_x/run.log
CENTERED=1 timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_l -- python3 $R/tools/perf_gemm.py 250000 100000 1 5 > $R/gpurun_out/prof_l/run.log 2>&1
grep mode= $R/gpurun_out/prof_l/run.log
