#!/usr/bin/env python3
"""random shapes / widths / engines / centring / spans / one- or two-pointer staging through the C ABI against the long-double dense oracle (tests/_util.Oracle):
fuzz_shapes.py [cases] [seed]      (MXA_SINGLE_ORIENTATION=1 in the environment runs the same through one-copy objects)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import miraculix_amd as mx
from _util import Oracle, make_problem

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 150
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
mx.load_shared_library()
dg = mx.dgemm_compressed
o = Oracle()
worst = 0.0
paths = {}
for c in range(cases):
    snps = int(rng.choice([rng.integers(1, 130), rng.integers(130, 700), rng.integers(700, 6000)]))
    indiv = int(rng.choice([rng.integers(1, 40), rng.integers(40, 600), rng.integers(600, 3000)]))
    n = int(rng.choice([1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 15, 16, 31, 32, 33, 40]))
    eng = str(rng.choice(["f64", "f64", "i8-exact", "f64-strict"]))
    centered = int(rng.integers(0, 2))
    prob = make_problem(snps, indiv, n, seed=int(rng.integers(1 << 30)), missing_frac=float(rng.choice([0.0, 0.02])))
    dg.set_engine(eng)
    dg.set_options(use_gpu=True, not_center=not centered, verbose=0)
    one_pointer = bool(rng.integers(0, 3) == 0)                         # a third of the objects staged from the SNP-major matrix alone (the library transposes)
    obj = dg.init_compressed(prob["plink"], None if one_pointer else prob["plink_t"], snps, indiv, prob["f"], n)
    try:
        for trans in (0, 1):
            k, m = (indiv, snps) if trans else (snps, indiv)
            B = rng.standard_normal((n, k)) * 10.0 ** rng.uniform(-3, 3, size=(n, 1))
            kind = int(rng.integers(0, 5))
            if kind == 1:
                B *= 10.0 ** rng.uniform(-12, 0, size=(n, k))          # wide spans inside the columns
            elif kind == 2:
                B[rng.integers(0, n)] = 0.0                             # a zero column
            elif kind == 3:
                B = np.round(B * 100.0)                                  # integers
            ld = k + int(rng.integers(0, 5))
            Bp = np.full((n, ld), np.nan); Bp[:, :k] = B                # poisoned ld padding
            C = dg.dgemm_compressed_main(bool(trans), obj, np.asfortranarray(Bp.T)[:k], snps, indiv)
            ref = o.dgemm_dense(trans, prob, B, centered)[:, :m]
            # relative to the magnitude the sums are formed from (a centred result can cancel to exactly zero: indiv = 1 has z - 2f = 0)
            mag = o.dgemm_dense(trans, prob, np.abs(B), 0)[:, :m].max(axis=1) + (2.0 * np.abs(B).sum(axis=1) if centered else 0.0)
            scale = np.maximum(np.maximum(np.abs(ref).max(axis=1), 1e-3 * mag), 1e-300)
            err = float((np.abs(C.T - ref).max(axis=1) / scale).max())
            worst = max(worst, err)
            paths[dg.last_path()] = paths.get(dg.last_path(), 0) + 1
            if not err <= 1e-11:
                print(f"FAIL case {c}: snps={snps} indiv={indiv} n={n} engine={eng} centred={centered} trans={trans} kind={kind} err={err:.3e} path={dg.last_path()}", flush=True)
                sys.exit(1)
    finally:
        dg.free_compressed(obj)
    if (c + 1) % 50 == 0:
        print(f"fuzz: {c + 1} / {cases} cases, worst so far {worst:.2e}", flush=True)   # a silent run is taken for a hung one on the GPU box
dg.set_engine("f64")
print(f"fuzz: {cases} cases x 2 products ok, worst column-wise relative error {worst:.2e}, paths {paths}")
