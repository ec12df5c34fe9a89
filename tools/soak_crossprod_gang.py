#!/usr/bin/env python3
"""repeat the config-3 crossproduct (gang-synchronised kernel) and watch the kernel time and a checksum: soak_crossprod_gang.py [reps]"""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import miraculix_amd as mx
from bench import synth_plink_device
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
k, rows = 500000, 100000
dev = torch.device("cuda", 0)
L = mx.load_shared_library()
X = synth_plink_device(torch, rows, (k + 3) // 4, 7, dev)
M = torch.zeros((rows, rows), dtype=torch.float64, device=dev)
ts, sums = [], set()
for i in range(reps):
    L.mxa_profile_reset()
    mx.crossproduct.snp_crossprod(X, k, rows, is_snpmajor=False, is_plink_format=True, out=M)
    torch.cuda.synchronize()
    la, ms = ctypes.c_int(0), ctypes.c_double(0)
    L.mxa_profile_get(ctypes.byref(la), ctypes.byref(ms))
    ts.append(ms.value / max(1, la.value))
    sums.add(float(M[::997, ::991].sum()))            # exact integers: the same value every time
    if (i + 1) % 10 == 0:
        print(f"soak_crossprod_gang: {i + 1} / {reps}, kernel ms min {min(ts):.1f} median {sorted(ts)[len(ts)//2]:.1f} max {max(ts):.1f}, distinct checksums {len(sums)}", flush=True)
assert len(sums) == 1
sub = M[:512, :512].cpu()
assert torch.equal(sub, sub.t())
print("soak ok")
