#!/bin/bash
# round 3: MFMA-pipe utilisation and effective clock of the narrow k_gemm tiles (n = 4, 8) and of the wide one (n = 32) on 500k x 50k
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
PMC="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS GRBM_GUI_ACTIVE"
for n in 4 8 32; do
  O=$R/gpurun_out/r03/pmc_n$n; rm -rf $O; mkdir -p $O
  MXA_ENGINE=f64-strict timeout -k 10 300 rocprofv3 --pmc $PMC --kernel-trace --output-format csv -d $O -- python3 $R/tools/perf_gemm.py 500000 50000 $n 2 > $O/run.log 2>&1
  (cd $R && python3 tools/pmc_mfma_util.py gpurun_out/r03/pmc_n$n "k_gemm<" gpurun_out/r03/pmc_n$n/util.json)
done
