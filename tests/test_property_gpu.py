"""Randomised shapes through the C ABI against the oracle (hypothesis, derandomised so every run draws the same cases): ragged sizes
around every granularity of the implementation (4 genotypes per byte, 16 per K-step, 128 per slab, 256 / 512 rows per tile / row block,
32-column chunks, the column peel, n <= 2), padded leading dimensions, missing codes, centred and not -- the edge cases the reference's
own tests only touch at two sizes (tests/dgemm_compressed/test.jl:88-104, tests/crossproduct/test_grm.jl:114-157)."""
import ctypes

import numpy as np
import pytest
from hypothesis import HealthCheck, given, settings, strategies as st

from _util import Oracle, make_problem

pytestmark = pytest.mark.gpu

dims = st.sampled_from([1, 2, 3, 4, 5, 7, 15, 16, 17, 63, 64, 127, 128, 129, 255, 256, 257, 500, 511, 512, 513, 1000, 1025, 2049])
ns = st.sampled_from([1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 13, 31, 32, 33, 34, 35, 64, 65, 70])


@settings(max_examples=60, deadline=None, derandomize=True, suppress_health_check=list(HealthCheck))
@given(snps=dims, indiv=dims, n=ns, centered=st.booleans(), trans=st.booleans(), pad=st.sampled_from([0, 1, 5]), missing=st.booleans(),
       seed=st.integers(0, 10_000))
def test_dgemm_random_shapes_vs_oracle(snps, indiv, n, centered, trans, pad, missing, seed):
    import miraculix_amd as mx
    L = mx.load_shared_library()
    dg = mx.dgemm_compressed
    o = Oracle()
    prob = make_problem(snps, indiv, n, seed=seed, missing_frac=0.1 if missing else 0.0)
    k = indiv if trans else snps
    m = snps if trans else indiv
    ldb, ldc = k + pad, m + pad
    rng = np.random.default_rng(seed + 1)
    B = np.full((n, ldb), 1e300)
    B[:, :k] = rng.standard_normal((n, k)) * 10.0 ** rng.integers(-3, 4, size=(n, 1))
    dg.set_options(use_gpu=True, not_center=not centered, verbose=0)
    obj = dg.init_compressed(prob["plink"], prob["plink_t"], snps, indiv, prob["f"], max(1, n // 2))     # n > max_n: the workspace grows
    try:
        C = np.full((n, ldc), -777.0)
        L.dgemm_compressed(b"T" if trans else b"N", obj, n, B.ctypes.data_as(ctypes.c_void_p), ldb, C.ctypes.data_as(ctypes.c_void_p), ldc)
        assert L.mxa_last_error() == 0
    finally:
        dg.free_compressed(obj)
    ref = o.dgemm_dense(int(trans), prob, B, int(centered))[:, :m]
    scale = np.abs(ref).max(axis=1, keepdims=True)
    scale[scale == 0] = 1.0
    assert (np.abs(C[:, :m] - ref) / scale).max() <= 1e-11, (snps, indiv, n, centered, trans, pad)
    assert np.all(C[:, m:] == 0.0)


@settings(max_examples=40, deadline=None, derandomize=True, suppress_health_check=list(HealthCheck))
@given(k=st.sampled_from([1, 3, 4, 5, 63, 64, 65, 127, 128, 129, 255, 257, 1000, 4097]), rows=st.sampled_from([1, 2, 31, 32, 33, 255, 256, 257, 300, 513, 700]),
       plink=st.booleans(), engine=st.sampled_from([None, "i8"]), seed=st.integers(0, 10_000))
def test_crossprod_random_shapes_bit_exact(k, rows, plink, engine, seed):
    import os
    import miraculix_amd as mx
    mx.load_shared_library()
    o = Oracle()
    rng = np.random.default_rng(seed)
    X = rng.integers(0, 256, size=(rows, (k + 3) // 4), dtype=np.uint8)
    if k % 4:
        X[:, -1] &= (1 << (2 * (k % 4))) - 1
    old = os.environ.pop("MXA_XPROD_ENGINE", None)
    try:
        if engine:
            os.environ["MXA_XPROD_ENGINE"] = engine
        M = mx.crossproduct.snp_crossprod(X, k, rows, is_snpmajor=False, is_plink_format=plink)
    finally:
        os.environ.pop("MXA_XPROD_ENGINE", None)
        if old is not None:
            os.environ["MXA_XPROD_ENGINE"] = old
    assert np.array_equal(M, o.crossprod_i32(X, k, plink).astype(np.float64)), (k, rows, plink, engine)


@settings(max_examples=30, deadline=None, derandomize=True, suppress_health_check=list(HealthCheck))
@given(snps=st.sampled_from([5, 16, 127, 130, 513, 1025, 3001]), indiv=st.sampled_from([3, 64, 257, 700]), n=st.sampled_from([1, 2, 3, 6, 10, 33]),
       shards=st.sampled_from([2, 3, 5, 8]), centered=st.booleans(), trans=st.booleans(), pad=st.sampled_from([0, 3]), seed=st.integers(0, 1000))
def test_multi_object_random_shapes_vs_oracle(snps, indiv, n, shards, centered, trans, pad, seed):
    """SNP-sharded objects behind the C ABI (virtual shards on one GPU) at ragged sizes: blocks of fewer than 128 SNPs, more shards than SNP
    quadruples, the small-n route and the column peel inside shards, padded leading dimensions"""
    import os
    import miraculix_amd as mx
    L = mx.load_shared_library()
    dg = mx.dgemm_compressed
    o = Oracle()
    prob = make_problem(snps, indiv, n, seed=seed)
    k = indiv if trans else snps
    m = snps if trans else indiv
    ldb, ldc = k + pad, m + pad
    rng = np.random.default_rng(seed + 7)
    B = np.full((n, ldb), 1e300)
    B[:, :k] = rng.standard_normal((n, k))
    dg.set_options(use_gpu=True, not_center=not centered, verbose=0)
    os.environ["MIRACULIX_NUM_GPUS"] = str(shards)
    try:
        obj = dg.init_compressed(prob["plink"], prob["plink_t"], snps, indiv, prob["f"], n)
    finally:
        os.environ.pop("MIRACULIX_NUM_GPUS", None)
    try:
        C = np.full((n, ldc), -777.0)
        L.dgemm_compressed(b"T" if trans else b"N", obj, n, B.ctypes.data_as(ctypes.c_void_p), ldb, C.ctypes.data_as(ctypes.c_void_p), ldc)
        assert L.mxa_last_error() == 0
    finally:
        dg.free_compressed(obj)
    ref = o.dgemm_dense(int(trans), prob, B, int(centered))[:, :m]
    scale = np.abs(ref).max(axis=1, keepdims=True)
    scale[scale == 0] = 1.0
    assert (np.abs(C[:, :m] - ref) / scale).max() <= 1e-11, (snps, indiv, n, shards, centered, trans, pad)
    assert np.all(C[:, m:] == 0.0)
