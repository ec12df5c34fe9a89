"""Opt-in engine 'i8-exact' (include/miraculix_amd.h, mxa_set_engine(4)): the exact int8 slicing for every n with the digit count chosen
per call from the measured exponent span of B's columns, so that B is represented WITHOUT error; the result must then obey the same
element-wise bound as the default engine's n <= 2 route -- |error| <= 3.02 (S - 1) 2^-53 sum_k |z_k b_k|, S <= 24 -- on adversarial
inputs (genotype rows that are zero exactly where B is large).  Beyond 24 digits, for inf / NaN, near the underflow threshold and for
K < 128 the fp64 MFMA path must run.  Oracle: long-double dense products (oracle/oracle.c)."""
import ctypes

import numpy as np
import pytest

from _util import Oracle, make_B, make_problem
from test_small_n_gpu import _adversarial_problem, _wide_B

pytestmark = pytest.mark.gpu
U = 2.0 ** -53


@pytest.fixture(autouse=True)
def _both_copies():
    """the opt-in engines at wide n multiply 'N' on the plain int8 kernel, which reads the individual-major copy: the objects of this module store BOTH
    copies (MXA_SINGLE_ORIENTATION=0; a default object keeps the SNP-major copy alone and sends such an 'N' to the fp64 engine)"""
    import os
    os.environ["MXA_SINGLE_ORIENTATION"] = "0"
    yield
    os.environ.pop("MXA_SINGLE_ORIENTATION", None)


@pytest.fixture(scope="module")
def mx():
    import miraculix_amd as m
    m.load_shared_library()
    prev = m.dgemm_compressed.set_engine("i8-exact")
    yield m
    m.dgemm_compressed.set_engine(prev)


def _run(mx, obj, prob, trans, B):
    return mx.dgemm_compressed.dgemm_compressed_main(bool(trans), obj, np.asfortranarray(B.T), prob["snps"], prob["indiv"]).T   # n x m


def _digits(mx):
    """digit count of the last int8 product (reported in the geometry's `a` field)"""
    L = mx.check_library_handle()
    gm, gk, gn, gs, ga, gc = ctypes.c_long(), ctypes.c_long(), ctypes.c_int(), ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    L.mxa_last_geometry(ctypes.byref(gm), ctypes.byref(gk), ctypes.byref(gn), ctypes.byref(gs), ctypes.byref(ga), ctypes.byref(gc))
    return ga.value


@pytest.mark.parametrize("n,decades", [(3, 6), (4, 12), (5, 20), (8, 3), (10, 30), (17, 9), (32, 6), (33, 16), (70, 5)])
@pytest.mark.parametrize("trans", [0, 1])
def test_obeys_the_stated_bound_elementwise(mx, n, decades, trans):
    o = Oracle()
    snps, indiv = 3001, 1037
    prob = _adversarial_problem(snps, indiv, seed=11)
    dg = mx.dgemm_compressed
    dg.set_options(use_gpu=True, not_center=True, verbose=0)
    obj = dg.init_compressed(prob["plink"], prob["plink_t"], snps, indiv, prob["f"], n)
    try:
        k = indiv if trans else snps
        m = snps if trans else indiv
        B = _wide_B(k, n, decades, seed=3 + trans, big_every=5 if trans else 7)
        C = _run(mx, obj, prob, trans, B)
        assert dg.last_path() == "k_gemm_i8"
        S = _digits(mx)
        span = max(np.frexp(np.abs(b).max())[1] - np.frexp(np.abs(b[b != 0]).min())[1] for b in B)
        assert S == max(7, -(-(span + 55) // 8)) and S <= 24
        ref = o.dgemm_dense(trans, prob, B, 0)[:, :m]
        abssum = o.dgemm_dense(trans, prob, np.abs(B), 0)[:, :m]           # sum_k |z_k b_k| per output
        err = np.abs(C - ref)
        bound = 3.02 * (S - 1) * U * abssum
        assert np.all(err <= bound + 1e-300), float((err / np.maximum(bound, 1e-300)).max())
        assert np.all(bound <= k * U * abssum)                              # below the bound of an fp64 FMA chain of this length
        small = abssum < 1e-3 * abssum.max()
        if small.any():
            assert np.all(err[small] <= 1e-13 * abssum[small])
        # centred, and bitwise repeatable
        dg.set_options(use_gpu=True, not_center=False, verbose=0)
        Cc = _run(mx, obj, prob, trans, B)
        assert dg.last_path() == "k_gemm_i8"
        refc = o.dgemm_dense(trans, prob, B, 1)[:, :m]
        assert np.abs(Cc - refc).max() <= 1e-11 * np.abs(refc).max()
        assert np.array_equal(Cc, _run(mx, obj, prob, trans, B))
    finally:
        dg.free_compressed(obj)


@pytest.mark.parametrize("n,decades", [(8, 3), (10, 30), (17, 9), (32, 6), (33, 16), (70, 5)])
def test_one_copy_objects_N_in_column_chunks_obeys_the_same_bound(mx, n, decades):
    """default objects keep ONE packed copy: 'N' at wide n runs the transposed-operand int8 kernel in column chunks of at most six digit tiles -- same digit
    count, same exact integer sums, same element-wise bound as the plain kernel"""
    import os
    os.environ["MXA_SINGLE_ORIENTATION"] = "1"
    o = Oracle()
    snps, indiv = 3001, 1037
    prob = _adversarial_problem(snps, indiv, seed=11)
    dg = mx.dgemm_compressed
    dg.set_options(use_gpu=True, not_center=True, verbose=0)
    obj = dg.init_compressed(prob["plink"], prob["plink_t"], snps, indiv, prob["f"], n)
    try:
        assert dg.single_orientation(obj) == 1
        B = _wide_B(snps, n, decades, seed=3, big_every=7)
        C = _run(mx, obj, prob, 0, B)
        assert dg.last_path() == "k_gemm_i8"
        S = _digits(mx)
        span = max(np.frexp(np.abs(b).max())[1] - np.frexp(np.abs(b[b != 0]).min())[1] for b in B)
        assert S == max(7, -(-(span + 55) // 8)) and S <= 24
        ref = o.dgemm_dense(0, prob, B, 0)[:, :indiv]
        abssum = o.dgemm_dense(0, prob, np.abs(B), 0)[:, :indiv]
        err = np.abs(C - ref)
        bound = 3.02 * (S - 1) * U * abssum
        assert np.all(err <= bound + 1e-300), float((err / np.maximum(bound, 1e-300)).max())
        dg.set_options(use_gpu=True, not_center=False, verbose=0)
        Cc = _run(mx, obj, prob, 0, B)
        assert dg.last_path() == "k_gemm_i8"
        refc = o.dgemm_dense(0, prob, B, 1)[:, :indiv]
        assert np.abs(Cc - refc).max() <= 1e-11 * np.abs(refc).max()
        assert np.array_equal(Cc, _run(mx, obj, prob, 0, B))
    finally:
        dg.free_compressed(obj)


@pytest.mark.parametrize("case", ["span", "inf", "nan", "tiny", "short_k"])
@pytest.mark.parametrize("n", [3, 12])
def test_declines_and_the_fp64_path_takes_over(mx, case, n):
    o = Oracle()
    snps, indiv = (100, 333) if case == "short_k" else (1500, 640)
    prob = make_problem(snps, indiv, n, seed=5)
    dg = mx.dgemm_compressed
    dg.set_options(use_gpu=True, not_center=True, verbose=0)
    obj = dg.init_compressed(prob["plink"], prob["plink_t"], snps, indiv, prob["f"], n)
    try:
        B = np.random.default_rng(6).standard_normal((n, snps))
        if case == "span":
            B[1, 3] = 1e-45       # 150 binades below the column maximum: more than 24 digits
        elif case == "inf":
            B[n - 1, 7] = np.inf
        elif case == "nan":
            B[0, 0] = np.nan
        elif case == "tiny":
            B[2] *= 1e-300        # the recombination would leave the normal range
        C = _run(mx, obj, prob, 0, B)
        assert dg.last_path() == "k_gemm"
        ref = o.dgemm_dense(0, prob, np.nan_to_num(B, nan=0.0, posinf=0.0), 0)[:, :indiv]
        for j in range(n):
            if np.isfinite(B[j]).all():
                assert np.abs(C[j] - ref[j]).max() <= 1e-11 * max(np.abs(ref[j]).max(), 1e-300), (case, j)
            else:
                assert not np.isfinite(C[j]).all()
    finally:
        dg.free_compressed(obj)


def test_integer_B_is_exact_and_matches_default_engine_closely(mx):
    rng = np.random.default_rng(3)
    prob = make_problem(3000, 400, 6, seed=11)
    dg = mx.dgemm_compressed
    dg.set_options(use_gpu=True, not_center=True, verbose=0)
    obj = dg.init_compressed(prob["plink"], prob["plink_t"], 3000, 400, prob["f"], 6)
    try:
        B = rng.integers(-(2 ** 20), 2 ** 20, size=(6, 3000)).astype(np.float64)
        C = _run(mx, obj, prob, 0, B)
        assert dg.last_path() == "k_gemm_i8"
        assert np.array_equal(C.T, (prob["Z"].astype(np.int64) @ B.T.astype(np.int64)).astype(np.float64))
        B = make_B(3000, 6, seed=2)[:, :3000]
        C8 = _run(mx, obj, prob, 0, B)
        assert dg.set_engine("f64-strict") == "i8-exact"
        C64 = _run(mx, obj, prob, 0, B)
        assert dg.last_path() == "k_gemm"
        assert dg.set_engine("i8-exact") == "f64-strict"
        assert np.abs(C8 - C64).max() <= 1e-12 * np.abs(C64).max()
    finally:
        dg.free_compressed(obj)


@pytest.mark.parametrize("copies,n", [(2, 12), (1, 12), (1, 40)])
def test_sharded_object_and_fused_gram_step(mx, copies, n):
    """the engine behind a multi-shard object (every shard chooses its own digit count for its rows of B) and through the fused CG step
    mxa_gram_matvec: against the oracle, and the fused step bit-identical to its two products.  copies = 1: default objects ('N' in column chunks on the
    transposed-operand kernel, per shard)"""
    import os
    os.environ["MXA_SINGLE_ORIENTATION"] = "1" if copies == 1 else "0"
    o = Oracle()
    snps, indiv = 4100, 901
    prob = make_problem(snps, indiv, n, seed=21)
    dg = mx.dgemm_compressed
    dg.set_options(use_gpu=True, not_center=False, verbose=0)
    old = os.environ.get("MIRACULIX_NUM_GPUS")
    os.environ["MIRACULIX_NUM_GPUS"] = "3"
    try:
        obj = dg.init_compressed(prob["plink"], prob["plink_t"], snps, indiv, prob["f"], n)
    finally:
        if old is None:
            os.environ.pop("MIRACULIX_NUM_GPUS", None)
        else:
            os.environ["MIRACULIX_NUM_GPUS"] = old
    try:
        assert dg.num_shards(obj) == 3
        for trans in (0, 1):
            k, m = (indiv, snps) if trans else (snps, indiv)
            B = make_B(k, n, seed=5 + trans)[:, :k]
            B[3] *= 1e-7
            C = _run(mx, obj, prob, trans, B)
            ref = o.dgemm_dense(trans, prob, B, 1)[:, :m]
            assert (np.abs(C - ref).max(axis=1) / np.abs(ref).max(axis=1)).max() <= 1e-11
    finally:
        dg.free_compressed(obj)
    obj = dg.init_compressed(prob["plink"], prob["plink_t"], snps, indiv, prob["f"], n)
    try:
        V = np.asfortranarray(make_B(indiv, n, seed=9)[:, :indiv].T)
        G = dg.gram_matvec(obj, V, snps, indiv)
        assert dg.last_path() == "k_gemm_i8"
        T = dg.dgemm_compressed_main(True, obj, V, snps, indiv)
        N = dg.dgemm_compressed_main(False, obj, np.asfortranarray(T), snps, indiv)
        assert np.array_equal(G, N)
        refT = o.dgemm_dense(1, prob, np.ascontiguousarray(V.T), 1)[:, :snps]
        refG = o.dgemm_dense(0, prob, refT, 1)[:, :indiv]
        assert np.abs(G.T - refG).max() <= 1e-10 * np.abs(refG).max()
    finally:
        dg.free_compressed(obj)
