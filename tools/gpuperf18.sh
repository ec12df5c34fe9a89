cd $GRAFT_REPO_ROOT
for u in 1536 3072 6144 12288; do MXA_LUT_UNITS=$u timeout -k 10 600 python tools/perf_gemm.py 250000 100000 1 5 2>&1 | grep -E "mode=|rror" | sed "s/^/units=$u /" | cut -c1-140; done
