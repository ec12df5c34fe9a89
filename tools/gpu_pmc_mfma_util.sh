cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/pmc4; mkdir -p $R/gpurun_out/pmc4
timeout -k 10 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/pmc4 -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-alt-engine > $R/gpurun_out/pmc4/run.log 2>&1
grep '^{' $R/gpurun_out/pmc4/run.log | cut -c1-120
