#!/bin/bash
# 2-bit transpose: parity tests, then kernel time of the tiled butterfly kernel against the generic one at 1M x 50k
cd "$GRAFT_REPO_ROOT"
timeout -k 10 600 python -m pytest tests/test_staging_gpu.py tests/test_largegrid_gpu.py -x -q 2>&1 | tail -3
cat > /tmp/tr.py <<'PY'
import os, sys, time, torch
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import miraculix_amd as mx
from bench import synth_plink_device
L = mx.load_shared_library()
snps, indiv = 1_000_000, 50_000
P = synth_plink_device(torch, snps, (indiv + 3) // 4, 1, torch.device("cuda", 0))
T = torch.empty((indiv, (snps + 3) // 4), dtype=torch.uint8, device="cuda")
for i in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    assert L.mxa_transpose_2bit(mx.lib.ptr(P), snps, indiv, mx.lib.ptr(T)) == 0
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"{'generic' if os.environ.get('MXA_TRANSPOSE_GENERIC') else 'tiled  '} transpose {snps} x {indiv}: {dt*1e3:.2f} ms = {2*P.numel()/dt/1e12:.2f} TB/s (read + write)")
print("checksum", int(T[:1000].sum()), int(T[-1000:].sum()))
PY
timeout -k 10 300 python /tmp/tr.py 2>&1 | grep -v amdgpu
MXA_TRANSPOSE_GENERIC=1 timeout -k 10 300 python /tmp/tr.py 2>&1 | grep -v amdgpu
