#!/bin/bash
# SQ counter passes for the MFMA utilisation of k_gemm (bench.py), k_gemm_i8 (opt-in engine) and k_crossprod2 (config-3-like shape)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
PMC="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS GRBM_GUI_ACTIVE"
rm -rf $R/gpurun_out/pmc_u1 $R/gpurun_out/pmc_u2 $R/gpurun_out/pmc_u3; mkdir -p $R/gpurun_out/pmc_u1 $R/gpurun_out/pmc_u2 $R/gpurun_out/pmc_u3
timeout -k 10 600 rocprofv3 --pmc $PMC --kernel-trace --output-format csv -d $R/gpurun_out/pmc_u1 -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-alt-engine > $R/gpurun_out/pmc_u1/run.log 2>&1
MXA_ENGINE=i8 timeout -k 10 600 rocprofv3 --pmc $PMC --kernel-trace --output-format csv -d $R/gpurun_out/pmc_u2 -- python3 $R/tools/perf_gemm.py 1000000 50000 32 1 > $R/gpurun_out/pmc_u2/run.log 2>&1
timeout -k 10 600 rocprofv3 --pmc $PMC --kernel-trace --output-format csv -d $R/gpurun_out/pmc_u3 -- python3 $R/tools/perf_crossprod.py 500000 49152 1 > $R/gpurun_out/pmc_u3/run.log 2>&1
cd $R
python3 tools/pmc_mfma_util.py gpurun_out/pmc_u1 "k_gemm<" gpurun_out/pmc_u1/util.json
python3 tools/pmc_mfma_util.py gpurun_out/pmc_u2 "k_gemm_i8" gpurun_out/pmc_u2/util.json
python3 tools/pmc_mfma_util.py gpurun_out/pmc_u3 "k_crossprod2" gpurun_out/pmc_u3/util.json
