"""Edge cases of the C ABI on the GPU: tiny and ragged dimensions, n larger than max_n (workspace growth), wide n that
spans several column chunks, repeated and interleaved calls on one object, two objects alive at once, switching centring
between calls, n = 0."""
import ctypes

import numpy as np
import pytest

from _util import Oracle, make_B, make_problem

pytestmark = pytest.mark.gpu
RTOL = 1e-11


@pytest.fixture(scope="module")
def mx():
    import miraculix_amd as m
    m.load_shared_library()
    return m


def _check(mx, o, obj, prob, trans, n, centered, seed=1):
    k = prob["indiv"] if trans else prob["snps"]
    m = prob["snps"] if trans else prob["indiv"]
    B = make_B(k, n, seed=seed)
    C = mx.dgemm_compressed.dgemm_compressed_main(bool(trans), obj, np.asfortranarray(B.T), prob["snps"], prob["indiv"])
    ref = o.dgemm_dense(trans, prob, B, centered)[:, :m]
    scale = max(np.abs(ref).max(), 1e-300)
    assert np.abs(C.T - ref).max() <= RTOL * scale, (trans, n)


@pytest.mark.parametrize("snps,indiv", [(1, 1), (3, 2), (5, 7), (127, 129), (128, 128), (129, 127), (513, 3), (2, 600)])
def test_tiny_and_ragged_dims(mx, snps, indiv):
    o = Oracle()
    prob = make_problem(snps, indiv, 1, seed=snps * 31 + indiv)
    dg = mx.dgemm_compressed
    dg.set_options(use_gpu=True, not_center=False, verbose=0)
    obj = dg.init_compressed(prob["plink"], prob["plink_t"], snps, indiv, prob["f"], 3)
    for trans in (0, 1):
        for n in (1, 3, 9):
            _check(mx, o, obj, prob, trans, n, 1)
    dg.free_compressed(obj)


def test_workspace_growth_and_wide_n(mx):
    o = Oracle()
    prob = make_problem(700, 300, 1, seed=4)
    dg = mx.dgemm_compressed
    dg.set_options(use_gpu=True, not_center=False, verbose=0)
    obj = dg.init_compressed(prob["plink"], prob["plink_t"], 700, 300, prob["f"], 1)   # max_n = 1
    for n in (1, 2, 17, 33, 100, 5):                                                   # n > max_n grows buffers
        for trans in (0, 1):
            _check(mx, o, obj, prob, trans, n, 1, seed=n)
    dg.free_compressed(obj)


def test_two_objects_and_option_switch(mx):
    o = Oracle()
    pa, pb = make_problem(900, 400, 1, seed=1), make_problem(333, 1200, 1, seed=2)
    dg = mx.dgemm_compressed
    dg.set_options(use_gpu=True, not_center=False, verbose=0)
    oa = dg.init_compressed(pa["plink"], pa["plink_t"], 900, 400, pa["f"], 8)
    ob = dg.init_compressed(pb["plink"], pb["plink_t"], 333, 1200, pb["f"], 8)
    for rep in range(3):
        _check(mx, o, oa, pa, rep % 2, 8, 1, seed=rep)
        _check(mx, o, ob, pb, (rep + 1) % 2, 4, 1, seed=rep + 10)
    dg.set_options(use_gpu=True, not_center=True, verbose=0)     # centring is a process-global option read at call time
    _check(mx, o, oa, pa, 0, 8, 0)
    _check(mx, o, ob, pb, 1, 2, 0)
    dg.free_compressed(oa)
    _check(mx, o, ob, pb, 0, 1, 0)
    dg.free_compressed(ob)


def test_n_zero_and_double_free(mx):
    prob = make_problem(100, 50, 1, seed=3)
    dg = mx.dgemm_compressed
    L = mx.check_library_handle()
    dg.set_options(use_gpu=True, not_center=True, verbose=0)
    obj = dg.init_compressed(prob["plink"], prob["plink_t"], 100, 50, prob["f"], 1)
    L.dgemm_compressed(b"N", obj, 0, None, 100, None, 50)        # n = 0: nothing to do, no error
    dg.free_compressed(obj)
    assert obj.value is None
    L.free_compressed(ctypes.byref(obj))                          # second free of a NULLed handle is a no-op
    with pytest.raises(RuntimeError):
        dg.free_compressed(obj)                                   # the binding layer reports the uninitialised pointer (tests/solve/test.jl:129)


def test_uncentred_without_frequencies(mx):
    """f may be omitted when centring is off (docs/genotype_matrix_multiplication.md: 'can be omitted with the options above')"""
    o = Oracle()
    prob = make_problem(640, 210, 1, seed=6)
    L = mx.check_library_handle()
    mx.dgemm_compressed.set_options(use_gpu=True, not_center=True, verbose=0)
    obj = ctypes.c_void_p(None)
    L.plink2compressed(prob["plink"].ctypes.data_as(ctypes.c_void_p), prob["plink_t"].ctypes.data_as(ctypes.c_void_p), 640, 210, None, 4, ctypes.byref(obj))
    assert obj.value
    _check(mx, o, obj, prob, 0, 4, 0)
    _check(mx, o, obj, prob, 1, 4, 0)
    mx.dgemm_compressed.free_compressed(obj)


def test_extreme_magnitudes_of_B(mx):
    """k_gemm feeds the genotypes as denormal doubles (z * 2^-1074) and scales every column of B by a power of two into a fixed
    window; the scaling must be invisible: columns at 1e-300, 1e+250, around the denormal threshold, an all-zero column, and a
    column that mixes 1e+20 with 1e-20 come out within the stated tolerance of the oracle (relative to each column's largest entry);
    non-finite input stays non-finite."""
    o = Oracle()
    snps, indiv, n = 2051, 700, 7
    prob = make_problem(snps, indiv, n, seed=77, missing_frac=0.03)
    dg = mx.dgemm_compressed
    dg.set_options(use_gpu=True, not_center=True, verbose=0)
    obj = dg.init_compressed(prob["plink"], prob["plink_t"], snps, indiv, prob["f"], n)
    try:
        for trans in (0, 1):
            k = indiv if trans else snps
            m = snps if trans else indiv
            B = make_B(k, n, seed=3)
            B[0] *= 1e-300
            B[1] *= 1e250
            B[2] *= 3e-308           # products with z = 2 straddle the smallest normal number
            B[3] = 0.0
            B[4, ::2] *= 1e20
            B[4, 1::2] *= 1e-20
            B[5] *= 2.0 ** -1040     # denormal inputs
            ref = o.dgemm_dense(trans, prob, B, 0)[:, :m]
            C = dg.dgemm_compressed_main(bool(trans), obj, np.asfortranarray(B.T), snps, indiv)
            assert np.all(np.isfinite(C))
            assert np.all(C[:, 3] == 0.0)
            for j in range(n):
                scale = np.abs(ref[j]).max()
                if scale > 0:
                    assert np.abs(C[:, j] - ref[j]).max() <= RTOL * scale, (trans, j)
            Binf = make_B(k, 5, seed=4)
            Binf[1, 5] = np.inf
            Cinf = dg.dgemm_compressed_main(bool(trans), obj, np.asfortranarray(Binf.T), snps, indiv)
            assert np.all(np.isfinite(Cinf[:, [0, 2, 3, 4]]))      # the other columns are untouched by the bad one
            assert not np.any(np.isfinite(Cinf[:, 1]))             # inf * z is inf (z > 0) or NaN (z = 0), as in plain fp64
    finally:
        dg.free_compressed(obj)


def _problem_from_Z(Z):
    from _util import pack_plink
    indiv, snps = Z.shape
    return dict(snps=snps, indiv=indiv, plink=np.ascontiguousarray(pack_plink(Z.T.copy())), plink_t=np.ascontiguousarray(pack_plink(Z)),
                f=Z.astype(np.float64).mean(axis=0) / 2.0, Z=Z)


@pytest.mark.parametrize("decades,expect_fallback", [(150, 1), (100, 0)])
def test_column_span_around_the_denormal_window(mx, decades, expect_fallback):
    """The fp64 MFMA path scales every column of B into a fixed window (genotype operand = the denormal z * 2^-1074); entries more than
    ~848 binades below their column's largest one would produce sub-normal products.  Adversarial input: a column that is 10^+decades at
    a few positions and 10^-decades * N(0,1) elsewhere, and output rows whose genotypes are ZERO exactly where the column is huge --
    their results are sums of the tiny entries only and a plain fp64 FMA chain (the reference's arithmetic,
    dgemm_compressed_cuda.h:259-266) gets them right.  decades = 150 (span 996 binades > kDenMaxSpan = 800): the per-call guard must send
    the product to the plain-operand fallback; decades = 100 (664 binades): inside the window, no fallback.  Element-wise against the
    long-double oracle either way."""
    from _util import synth_genotypes
    o = Oracle()
    snps, indiv, n = 1500, 600, 8   # n = 8: the fp64 MFMA path (n <= 6 takes the guarded int8 route, whose fallback is the fp64 chain kernel)
    Z, _ = synth_genotypes(snps, indiv, seed=5)
    Z[:12, :48] = 0          # 'N': individuals 0..11 carry no allele at SNPs 0..47
    Z[:40, :9] = 0           # 'T': SNPs 0..8 are zero in individuals 0..39
    prob = _problem_from_Z(Z)
    dg = mx.dgemm_compressed
    L = mx.check_library_handle()
    dg.set_options(use_gpu=True, not_center=True, verbose=0)
    obj = dg.init_compressed(prob["plink"], prob["plink_t"], snps, indiv, prob["f"], n)
    try:
        for trans, nbig, quiet in ((0, 48, slice(0, 12)), (1, 40, slice(0, 9))):
            k = indiv if trans else snps
            m = snps if trans else indiv
            B = make_B(k, n, seed=8)
            B[2, :nbig] = 10.0 ** decades * (1.0 + np.arange(nbig) / 64.0)
            B[2, nbig:] *= 10.0 ** -decades
            ref = o.dgemm_dense(trans, prob, B, 0)[:, :m]
            C = dg.dgemm_compressed_main(bool(trans), obj, np.asfortranarray(B.T), snps, indiv)
            assert L.mxa_last_range_fallback(obj) == expect_fallback
            for j in range(n):
                assert np.abs(C[:, j] - ref[j]).max() <= RTOL * np.abs(ref[j]).max(), (trans, j)
            q_ref, q_got = ref[2][quiet], C[quiet, 2]
            assert np.all(q_ref != 0.0) and np.abs(q_ref).max() < 10.0 ** (3 - decades)       # tiny, but not nothing
            assert np.abs(q_got - q_ref).max() <= 1e-12 * np.abs(q_ref).max(), (trans, q_got, q_ref)
            # the next ordinary product on the same object is back on the fast path
            B2 = make_B(k, n, seed=9)
            C2 = dg.dgemm_compressed_main(bool(trans), obj, np.asfortranarray(B2.T), snps, indiv)
            assert L.mxa_last_range_fallback(obj) == 0
            ref2 = o.dgemm_dense(trans, prob, B2, 0)[:, :m]
            assert np.abs(C2.T - ref2).max() <= RTOL * np.abs(ref2).max()
    finally:
        dg.free_compressed(obj)
