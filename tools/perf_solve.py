#!/usr/bin/env python3
"""Solver twin timing at the reference test's largest size (tests/solve/test.jl: n = 15e3, ncol 20): dense Cholesky solve and the
two sparse triangular solves.  usage: perf_solve.py [n] [ncol]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, scipy.sparse
import miraculix_amd as mx

n = int(sys.argv[1]) if len(sys.argv) > 1 else 15000
ncol = int(sys.argv[2]) if len(sys.argv) > 2 else 20
mx.load_shared_library()
rng = np.random.default_rng(0)
idx = np.arange(n, dtype=np.float64)
M = np.exp(-np.abs(idx[:, None] - idx[None, :]) / n) + 1e-3 * np.eye(n)
B = rng.standard_normal((n, ncol)) + 5.0
for rep in range(2):
    t0 = time.perf_counter(); X, ld = mx.solve.dense_solve(M, B); dt = time.perf_counter() - t0
    print(f"dense_solve n={n} ncol={ncol}: {dt:.2f} s ({'first call' if rep == 0 else 'warm'}), residual {np.linalg.norm(M @ X - B)/np.linalg.norm(B):.1e}, logdet {ld:.6f}", flush=True)
nnz_per_row = 20
rows = np.repeat(np.arange(n), nnz_per_row); cols = rng.integers(0, n, size=n * nnz_per_row)
keep = cols > rows
U = scipy.sparse.coo_matrix((rng.random(keep.sum()) * 0.04, (rows[keep], cols[keep])), shape=(n, n)).tocsr()
U.sum_duplicates()
U = (U + scipy.sparse.diags(np.maximum(rng.standard_normal(n) + 2.0, 1.0))).tocoo()
obj = mx.solve.sparse_init(U.data, (U.row + 1).astype(np.int64), (U.col + 1).astype(np.int64), U.nnz, n, ncol, False)
for tr in ("t", "n"):
    t0 = time.perf_counter(); Y = mx.solve.sparse_solve(obj, tr, B, n); dt = time.perf_counter() - t0
    Ucsr = U.tocsr()
    R = (Ucsr.T @ Y if tr == "t" else Ucsr @ Y) - B
    print(f"sparse_solve '{tr}' n={n} nnz={U.nnz} ncol={ncol}: {dt*1e3:.1f} ms, residual {np.linalg.norm(R)/np.linalg.norm(B):.1e}", flush=True)
mx.solve.sparse_free(obj)
