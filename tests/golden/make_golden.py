#!/usr/bin/env python3
"""Emit the golden fixtures tests/golden/dgemm_golden.npz from the REFERENCE's own CPU library.

Run in the build container only (needs /root/reference and `make -C oracle ref`, which compiles the reference's
sources where they lie into oracle/_ref/): the expected outputs stored here are what
setOptions_compressed(0,cores,0,0,1,!centre,0,0,256,0) -> plink2compressed -> dgemm_compressed of the reference return
(driver oracle/ref_driver.c).  Fixtures are data only: inputs (PLINK bytes, f, B) and the reference's outputs.
Shapes follow SURVEY.md 8c (non-multiple-of-4/5 dims, missings, external f, padded ldb/ldc) and avoid the individual
counts for which the reference's re-encoder hangs.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from _util import have_reference, make_B, make_problem, run_reference  # noqa: E402

CORES = 8
CASES = [
    # name, snps, indiv, n, missing_frac, perturb_f, ldb_pad, ldc_pad
    ("c1_1000x500_n1", 1000, 500, 1, 0.0, False, 0, 0),         # BASELINE.json configs[0]
    ("odd_1003x501_n5", 1003, 501, 5, 0.0, False, 0, 0),
    ("odd_1003x501_n7_ld", 1003, 501, 7, 0.0, False, 3, 5),
    ("n32_1003x501", 1003, 501, 32, 0.0, False, 0, 0),
    ("sq_2000x1000_n8", 2000, 1000, 8, 0.0, False, 0, 0),
    ("missing_1203x610_n6", 1203, 610, 6, 0.1, False, 0, 0),
    ("extf_777x1301_n10", 777, 1301, 10, 0.0, True, 0, 0),
]


def main():
    assert have_reference(), "build the reference first: make -C oracle ref"
    out = {}
    names = []
    for name, snps, indiv, n, miss, perturb, ldb_pad, ldc_pad in CASES:
        prob = make_problem(snps, indiv, n, seed=abs(hash(name)) % 10000 if False else sum(map(ord, name)), missing_frac=miss)
        if perturb:  # external frequencies that differ from the data: must be used verbatim (haplogeno.cc:1591-1593)
            rng = np.random.default_rng(5)
            prob["f"] = np.clip(prob["f"] + rng.uniform(-0.05, 0.05, size=snps), 0.0, 1.0)
        names.append(name)
        out[f"{name}/plink"] = prob["plink"]
        out[f"{name}/plink_t"] = prob["plink_t"]
        out[f"{name}/f"] = prob["f"]
        out[f"{name}/dims"] = np.array([snps, indiv, n, ldb_pad, ldc_pad], dtype=np.int64)
        for trans in (0, 1):
            k = indiv if trans else snps
            m = snps if trans else indiv
            ldb, ldc = k + ldb_pad, m + ldc_pad
            B = make_B(k, n, seed=43 + trans, ldb=ldb)
            if ldb_pad:
                B[:, k:] = 12345.0  # finite poison in the ld padding (must never be read as data)
            out[f"{name}/B{trans}"] = B
            for centered in (0, 1):
                C, _ = run_reference(prob, trans, B, centered, ldc=ldc, variant=256, cores=CORES)
                out[f"{name}/C{trans}{centered}"] = C
        print("done", name)
    out["names"] = np.array(names)
    out["cores"] = np.array([CORES])
    np.savez_compressed(os.path.join(HERE, "dgemm_golden.npz"), **out)
    print("wrote", os.path.join(HERE, "dgemm_golden.npz"), os.path.getsize(os.path.join(HERE, "dgemm_golden.npz")), "bytes")


if __name__ == "__main__":
    main()
