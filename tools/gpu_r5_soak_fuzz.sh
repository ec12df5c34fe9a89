#!/bin/bash
# round 5: soak and fuzz of the final code -- default objects (one packed copy) and MXA_SINGLE_ORIENTATION=0 (two copies); smoke()
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
{
echo "# tools/gpu_r5_soak_fuzz.sh on one MI355X, final code of round 5"
echo "== smoke"; timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
for so in 1 0; do
  echo "== MXA_SINGLE_ORIENTATION=$so: tools/soak.py 400"
  MXA_SINGLE_ORIENTATION=$so timeout -k 10 600 python tools/soak.py 400 2>&1 | tail -3
  echo "== MXA_SINGLE_ORIENTATION=$so: tools/fuzz_shapes.py 250"
  MXA_SINGLE_ORIENTATION=$so timeout -k 10 600 python tools/fuzz_shapes.py 250 $so 2>&1 | tail -3
done
echo "== tools/soak_lifecycle.py 40"; timeout -k 10 600 python tools/soak_lifecycle.py 40 2>&1 | tail -3
echo "== tools/fuzz_crossprod.py 120"; timeout -k 10 600 python tools/fuzz_crossprod.py 120 2>&1 | tail -3
} > gpurun_out/r5_soak_fuzz.txt 2>&1
cat gpurun_out/r5_soak_fuzz.txt
