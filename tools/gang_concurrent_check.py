import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import miraculix_amd as mx
from _util import pack_plink
mx.load_shared_library()
rng = np.random.default_rng(1)
rows, k = 12000, 140000
Z = rng.integers(0, 3, size=(rows, k)).astype(np.int8)
X = np.ascontiguousarray(pack_plink(Z))
t0 = time.time()
M = mx.crossproduct.snp_crossprod(X, k, rows, is_snpmajor=False, is_plink_format=True)
print("two concurrent gang kernels on one device: %.2f s" % (time.time() - t0))
idx = rng.integers(0, rows, size=200)
ref = (Z[idx].astype(np.int64) @ Z.astype(np.int64).T).astype(np.float64)
assert np.array_equal(M[idx], ref)
assert np.array_equal(M, M.T)
print("concurrent ok")
