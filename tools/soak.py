#!/usr/bin/env python3
"""Soak run: a few hundred calls through the C ABI with changing n, engine, centring and host/device operands on one object;
checks every result against a reference computed once and that device memory does not creep."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import miraculix_amd as mx
from _util import make_problem, make_B

mx.load_shared_library()
dg = mx.dgemm_compressed
snps, indiv = 150011, 2001   # B of 'N' / C of 'T' at n >= 28 exceed 32 MB: the host-operand pipeline takes part
prob = make_problem(snps, indiv, 1, seed=1, missing_frac=0.01)
Zc = prob["Z"].astype(np.float64)
rng = np.random.default_rng(0)
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 300
free0 = None
worst = 0.0
t0 = time.time()
dg.set_options(use_gpu=True, not_center=True, verbose=0)
obj = dg.init_compressed(prob["plink"], prob["plink_t"], snps, indiv, prob["f"], 4)
for it in range(iters):
    n = int(rng.choice([1, 2, 3, 4, 5, 6, 7, 8, 10, 11, 13, 15, 17, 32, 33, 34, 64, 130]))
    trans = bool(rng.integers(0, 2))
    eng = str(rng.choice(["f64", "f64", "f64", "i8", "f64-strict", "i8-exact", "i8-exact"]))
    dg.set_engine(eng)
    centered = bool(rng.integers(0, 2))
    dg.set_options(use_gpu=True, not_center=not centered, verbose=0)
    k = indiv if trans else snps
    B = rng.standard_normal((k, n))
    ref = (Zc.T @ B) if trans else (Zc @ B)
    if centered:
        ref = ref - 2.0 * (np.outer(prob["f"], B.sum(axis=0)) if trans else np.outer(np.ones(indiv), prob["f"] @ B))
    if rng.integers(0, 2):
        C = dg.dgemm_compressed_main(trans, obj, np.asfortranarray(B), snps, indiv)
    else:
        C = dg.dgemm_compressed_main(trans, obj, torch.from_numpy(np.ascontiguousarray(B.T)).cuda().t(), snps, indiv).cpu().numpy()
    err = np.abs(C - ref).max() / np.abs(ref).max()
    worst = max(worst, err)
    assert err < 1e-11, (it, n, trans, eng, centered, err)
    if it == iters // 2:
        torch.cuda.synchronize(); free0 = torch.cuda.mem_get_info()[0]
dg.set_engine("f64")
free1 = torch.cuda.mem_get_info()[0]
dg.free_compressed(obj)
print(f"soak: {iters} calls ok in {time.time()-t0:.1f} s, worst rel err {worst:.2e}, device memory drift over the second half of the calls (every shape and engine seen before): {(free0 - free1)/2**20:.1f} MiB")
