#!/bin/bash
# kernel stats of mxa_grm / mxa_ld fused vs unfused at config-3 size
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out/r03"; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
rm -rf "$O/tmp_grm"
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/tmp_grm" -- python3 "$R/tools/perf_grm.py" 500000 100000 > "$O/grm_prof_run.log" 2>&1
f=$(find "$O/tmp_grm" -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cut -c1-200 "$f" | grep "mxa::" > "$O/grm_kernel_stats.csv"
rm -rf "$O/tmp_grm"
cat "$O/grm_kernel_stats.csv" | cut -c1-60,150-200 ; grep -v amdgpu "$O/grm_prof_run.log" | tail -7
