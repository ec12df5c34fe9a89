#!/usr/bin/env python3
"""single-device object in a create / multiply / free loop; the 'T' result goes to a host matrix with ldc = m + 5 (strided download).
usage: leak_probe3.py [reuse]   -- reuse: one object, only the multiply loops"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import miraculix_amd as mx
from _util import make_problem
L = mx.load_shared_library()
dg = mx.dgemm_compressed
snps, indiv, n = 6001, 1201, 10
prob = make_problem(snps, indiv, n, seed=3)
dg.set_options(use_gpu=True, not_center=True, verbose=0)
BT = np.asfortranarray(np.random.default_rng(0).standard_normal((indiv, n)))
def free(): torch.cuda.synchronize(); return torch.cuda.mem_get_info()[0]
reuse = len(sys.argv) > 1 and sys.argv[1] == "reuse"
obj = dg.init_compressed(prob["plink"], prob["plink_t"], snps, indiv, prob["f"], n) if reuse else None
base = None
for it in range(110):
    o = obj if reuse else dg.init_compressed(prob["plink"], prob["plink_t"], snps, indiv, prob["f"], n)
    C = np.zeros((n, snps + 5))
    L.dgemm_compressed(b"T", o, n, BT.ctypes.data_as(ctypes.c_void_p), indiv, C.ctypes.data_as(ctypes.c_void_p), snps + 5)
    assert L.mxa_last_error() == 0
    if not reuse: dg.free_compressed(o)
    if it == 9: base = free()
print(f"leak probe 3 ({'one object' if reuse else 'object per iteration'}, MXA_COPY_COLUMNS={os.environ.get('MXA_COPY_COLUMNS','0')}): drift over 100 iterations {(base - free()) / 2**20:.1f} MiB")
