cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/pmc1
timeout -k 10 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $R/gpurun_out/pmc1 -- python3 $R/tools/perf_gemm.py 200000 50000 32 1 > $R/gpurun_out/pmc1/run.log 2>&1
tail -3 $R/gpurun_out/pmc1/run.log
ls -R $R/gpurun_out/pmc1 | head -20
