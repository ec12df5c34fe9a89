cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/pmc3; mkdir -p $R/gpurun_out/pmc3
CENTERED=0 timeout -k 10 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS --kernel-trace --output-format csv -d $R/gpurun_out/pmc3 -- python3 $R/tools/perf_gemm.py 250000 100000 1 1 > $R/gpurun_out/pmc3/run.log 2>&1
tail -2 $R/gpurun_out/pmc3/run.log
