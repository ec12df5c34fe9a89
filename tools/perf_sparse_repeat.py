#!/usr/bin/env python3
"""repeat the sparse triangular solves to look for outliers of the flag-polling kernel: perf_sparse_repeat.py [n] [ncol] [reps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, scipy.sparse
import miraculix_amd as mx
n = int(sys.argv[1]) if len(sys.argv) > 1 else 15000
ncol = int(sys.argv[2]) if len(sys.argv) > 2 else 20
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
mx.load_shared_library()
rng = np.random.default_rng(0)
B = rng.standard_normal((n, ncol)) + 5.0
rows = np.repeat(np.arange(n), 20); cols = rng.integers(0, n, size=n * 20)
keep = cols > rows
U = scipy.sparse.coo_matrix((rng.random(keep.sum()) * 0.04, (rows[keep], cols[keep])), shape=(n, n)).tocsr()
U.sum_duplicates()
U = (U + scipy.sparse.diags(np.maximum(rng.standard_normal(n) + 2.0, 1.0))).tocoo()
obj = mx.solve.sparse_init(U.data, (U.row + 1).astype(np.int64), (U.col + 1).astype(np.int64), U.nnz, n, ncol, False)
for tr in ("t", "n"):
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); Y = mx.solve.sparse_solve(obj, tr, B, n); ts.append((time.perf_counter() - t0) * 1e3)
    print(f"sparse_solve '{tr}' n={n} nnz={U.nnz} ncol={ncol}: ms", " ".join(f"{t:.1f}" for t in ts), flush=True)
mx.solve.sparse_free(obj)
