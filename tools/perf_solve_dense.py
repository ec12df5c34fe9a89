#!/usr/bin/env python3
"""dense Cholesky solve only (n, ncol from argv): for rocprofv3 --kernel-trace --stats"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
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
    print(f"dense_solve n={n} ncol={ncol}: {dt:.2f} s, residual {np.linalg.norm(M @ X - B)/np.linalg.norm(B):.1e}, logdet {ld:.6f}", flush=True)
