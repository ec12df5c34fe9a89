#!/usr/bin/env python3
"""kernel-level perf of snp_multiply_gpu on synthetic device data.  usage: perf_crossprod.py k(snps) rows(indiv) [reps]"""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import miraculix_amd as mx
from bench import synth_plink_device

k, rows = int(sys.argv[1]), int(sys.argv[2])
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 2
dev = torch.device("cuda", 0)
L = mx.load_shared_library()
X = synth_plink_device(torch, rows, (k + 3) // 4, 7, dev)
M = torch.zeros((rows, rows), dtype=torch.float64, device=dev)
mx.crossproduct.snp_crossprod(X, k, rows, is_snpmajor=False, is_plink_format=True, out=M)
L.mxa_profile_reset()
t0 = time.perf_counter()
for _ in range(reps):
    mx.crossproduct.snp_crossprod(X, k, rows, is_snpmajor=False, is_plink_format=True, out=M)
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / reps
la, ms = ctypes.c_int(0), ctypes.c_double(0)
L.mxa_profile_get(ctypes.byref(la), ctypes.byref(ms))
avg = ms.value / la.value
ops = 2.0 * k * rows * rows
print(f"crossprod k={k} rows={rows}: kernel {avg:.2f} ms = {ops/avg*1e-12:.2f} Pop/s (full-matrix count 2*k*rows^2; the kernel executes the upper triangle only); call wall {wall*1e3:.1f} ms", flush=True)
# spot check: symmetric, diagonal = sum z^2
sub = M[:4, :4].cpu()
assert torch.equal(sub, sub.t())
