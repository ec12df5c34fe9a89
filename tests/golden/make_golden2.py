#!/usr/bin/env python3
"""Second set of golden fixtures from the REFERENCE's own CPU library (tests/golden/dgemm_golden2.npz; same recipe and driver as make_golden.py,
which stays untouched so that its file does not change): the operand patterns of the reference's Fortran integration tests
(tests/dgemm_compressed/test_5codesapi.f90:244-248: 'n', ncol = 10, B(i, j) = -(10 i + j);  test_5codesapi_t.f90:221-225: 't', ncol = 15,
B(i, j) = -(1000 i + j)), wider products (n = 40 with missings, n = 65: three column chunks of the MFMA tile) and the reference's other
SIMD variants (32: plain C, 128: SSE) on one case.  Run in the build container only (needs /root/reference and `make -C oracle ref`)."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from _util import have_reference, make_B, make_problem, run_reference  # noqa: E402

CORES = 8
CASES = [
    # name, snps, indiv, n, missing_frac, B pattern, variant
    ("fortran_n_1201x603_n10", 1201, 603, 10, 0.0, "fortran", 256),
    ("fortran_t_1201x603_n15", 1201, 603, 15, 0.0, "fortran", 256),
    ("wide_705x503_n40_missing", 705, 503, 40, 0.05, "randn", 256),
    ("wide_600x403_n65", 600, 403, 65, 0.0, "randn", 256),
    ("variant32_1003x501_n9", 1003, 501, 9, 0.02, "randn", 32),
    ("variant128_1003x501_n9", 1003, 501, 9, 0.02, "randn", 128),
]


def fortran_B(k, n, trans):
    i = np.arange(1, k + 1, dtype=np.float64)[None, :]
    j = np.arange(1, n + 1, dtype=np.float64)[:, None]
    return -((1000.0 if trans else 10.0) * i + j)      # row j of the array = column j of the column-major matrix


def main():
    assert have_reference(), "build the reference first: make -C oracle ref"
    out, names = {}, []
    for name, snps, indiv, n, miss, pattern, variant in CASES:
        prob = make_problem(snps, indiv, n, seed=sum(map(ord, name)), missing_frac=miss)
        names.append(name)
        out[f"{name}/plink"] = prob["plink"]
        out[f"{name}/plink_t"] = prob["plink_t"]
        out[f"{name}/f"] = prob["f"]
        out[f"{name}/dims"] = np.array([snps, indiv, n, 0, 0], dtype=np.int64)
        for trans in (0, 1):
            k = indiv if trans else snps
            B = fortran_B(k, n, trans) if pattern == "fortran" else make_B(k, n, seed=143 + trans)
            out[f"{name}/B{trans}"] = B
            for centered in (0, 1):
                C, _ = run_reference(prob, trans, B, centered, variant=variant, cores=CORES)
                out[f"{name}/C{trans}{centered}"] = C
        print("done", name)
    out["names"] = np.array(names)
    out["cores"] = np.array([CORES])
    np.savez_compressed(os.path.join(HERE, "dgemm_golden2.npz"), **out)


if __name__ == "__main__":
    main()
