#!/bin/bash
# CPU-side sanitizer pass (reference practice: src/miraculix/makefile.c.mk:47-50 -fsanitize=address profile,
# tests/dgemm_compressed/Makefile:80-85 valgrind targets).  Never on the GPU build.
#  1. the library's host-only planners (miraculix_amd/csrc/mxa_plan.h) swept over shapes under ASan + UBSan (tests/host/plan_sweep.cpp);
#  2. the oracle (oracle/oracle.c) rebuilt with -fsanitize=address,undefined and driven by its own tests (tests/test_oracle.py, tests/test_sparse_cpu.py)
#     with libasan preloaded into the python process (leak detection off: the interpreter is not instrumented).
set -e
cd "$(dirname "$0")/.."
out=profiles/${MXA_ROUND:-r06}_sanitizers.txt
{
  echo "# tools/run_sanitizers.sh  ($(gcc --version | head -1))"
  echo "## 1. planners under ASan + UBSan"
  make -C tests/host clean >/dev/null; make -C tests/host san 2>&1 | tail -2
  echo "## 2. oracle under ASan + UBSan"
  make -C oracle san 2>&1 | tail -1
  ASAN=$(gcc -print-file-name=libasan.so); UBSAN=$(gcc -print-file-name=libubsan.so)
  LD_PRELOAD="$ASAN:$UBSAN" ASAN_OPTIONS=detect_leaks=0:abort_on_error=1 UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1 OMP_NUM_THREADS=4 \
    ORACLE_SO=$PWD/oracle/liboracle_san.so python -m pytest tests/test_oracle.py tests/test_sparse_cpu.py -x -q -p no:cacheprovider 2>&1 | tail -4
} | tee $out
