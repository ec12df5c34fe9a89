#!/bin/bash
# round 5: SQ counter pass of the CURRENT crossproduct kernel k_crossprod_gang<false,0> (FP4) and <true,0> (int8) at BASELINE config 3 (500k SNPs x 100k indiv):
# MFMA pipe busy, effective clock, wait fractions (the only counter pass under profiles/ was of the pre-gang kernel of round 2);
# then the call-wall outliers of device-result calls with the PRINT_LEVEL=1 phase clock
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
PMC="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS GRBM_GUI_ACTIVE"
for eng in f4 i8; do
  O=$R/gpurun_out/r05x_$eng; rm -rf $O; mkdir -p $O
  if [ $eng = i8 ]; then export MXA_XPROD_ENGINE=i8; else unset MXA_XPROD_ENGINE; fi
  timeout -k 10 500 rocprofv3 --pmc $PMC --kernel-trace --output-format csv -d $O -- python3 $R/tools/perf_crossprod.py 500000 100000 2 > $O/run.log 2>&1 || { tail -5 $O/run.log; exit 1; }
  tail -1 $O/run.log
  (cd $R && python3 tools/pmc_mfma_util.py $O "k_crossprod_gang" gpurun_out/r05_pmc_mfma_util_crossprod_gang_$eng.json)
  rm -rf $O/*/  # the raw counter csv files are large
done
unset MXA_XPROD_ENGINE
cd $R
{
echo "# device-result calls of snp_multiply_gpu at config 3, PRINT_LEVEL=1 phase clock (six calls in one process)"
PRINT_LEVEL=1 timeout -k 10 300 python3 tools/perf_crossprod.py 500000 100000 6 2>&1 | grep -v "^--\|miraculix_amd (MI\|Compiled on" | tail -60
} > gpurun_out/r05_crossprod_call_wall.txt 2>&1
tail -40 gpurun_out/r05_crossprod_call_wall.txt
