"""CPU: the oracle's sparse_times_plink restatement against the golden vectors captured from the reference library
(tests/golden/make_golden_sparse.py).  The reference sums the stored entries in groups of up to 8, the oracle in long double:
agreement to rounding."""
import os

import numpy as np

from _util import Oracle

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "sparse_golden.npz")
RTOL = 1e-13


def test_oracle_sparse_matches_reference_golden():
    g = np.load(GOLD)
    o = Oracle()
    for name in g["names"]:
        snps, indiv, ldc, nidx = (int(x) for x in g[f"{name}/dims"])
        for tc in ("N", "T"):
            P = g[f"{name}/plink_t"] if tc == "T" else g[f"{name}/plink"]
            rows, entries = (indiv, snps) if tc == "T" else (snps, indiv)
            C = o.sparse_times_plink(np.ascontiguousarray(P), rows, entries, g[f"{name}/ia{tc}"], g[f"{name}/ja{tc}"], g[f"{name}/a{tc}"], ldc=ldc)
            ref = g[f"{name}/C{tc}"]
            assert C.shape == ref.shape == (entries, ldc)
            assert np.abs(C - ref).max() <= RTOL * max(1.0, np.abs(ref).max()), (name, tc)
            assert np.all(ref[:, nidx:] == 0.0) and np.all(C[:, nidx:] == 0.0)   # ld padding rows are zero-filled
