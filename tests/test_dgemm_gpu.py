"""GPU parity tests proper: the HIP path, called through the C ABI, against the oracle on the same seeded inputs.
Mirrors tests/dgemm_compressed/test.jl:88-104 and test_5codesapi{,_t}.f90 of the reference (random B with n=10/15,
dense (G - 2f) B oracle, max-abs comparison) with a much tighter, stated tolerance."""
import numpy as np
import pytest

from _util import Oracle, make_B, make_problem

pytestmark = pytest.mark.gpu

# fp64 tolerance of the path (stated): max|C - C_ref| <= 1e-11 * max|C_ref|
RTOL = 1e-11


@pytest.fixture(scope="module")
def mx():
    import miraculix_amd as m
    m.load_shared_library()
    return m


def _run(mx, prob, trans, B, centered, ldc=None):
    dg = mx.dgemm_compressed
    dg.set_options(use_gpu=True, not_center=not centered, verbose=0)
    obj = dg.init_compressed(prob["plink"], prob["plink_t"], prob["snps"], prob["indiv"], prob["f"], B.shape[0])
    try:
        k = prob["indiv"] if trans else prob["snps"]
        Bcm = np.asfortranarray(B[:, :k].T)  # (k x n) column-major
        C = dg.dgemm_compressed_main(bool(trans), obj, Bcm, prob["snps"], prob["indiv"])
    finally:
        dg.free_compressed(obj)
    assert obj.value is None
    return C


@pytest.mark.parametrize("snps,indiv,n", [(1000, 500, 1), (1003, 501, 2), (2047, 771, 3), (5000, 1203, 4), (1003, 501, 5), (2000, 1000, 7), (777, 1301, 10), (4100, 515, 15), (3001, 2050, 32), (1500, 700, 40), (900, 1100, 65), (1200, 640, 128)])
@pytest.mark.parametrize("trans", [0, 1])
@pytest.mark.parametrize("centered", [0, 1])
def test_dgemm_vs_oracle(mx, snps, indiv, n, trans, centered):
    o = Oracle()
    prob = make_problem(snps, indiv, n, seed=42 + snps)
    k = indiv if trans else snps
    m = snps if trans else indiv
    B = make_B(k, n, seed=43)
    ref = o.dgemm_dense(trans, prob, B, centered)[:, :m]
    C = _run(mx, prob, trans, B, centered)
    assert C.shape == (m, n)
    err = np.abs(C.T - ref).max() / np.abs(ref).max()
    assert err <= RTOL, err


@pytest.mark.parametrize("snps,indiv,n", [(1000, 500, 1), (2047, 771, 3), (5000, 1203, 4), (1003, 501, 5), (2600, 900, 6), (777, 1301, 10)])
@pytest.mark.parametrize("trans", [0, 1])
@pytest.mark.parametrize("centered", [0, 1])
def test_dgemm_vs_oracle_fp64_arithmetic_only(mx, snps, indiv, n, trans, centered):
    """engine 'f64-strict': fp64 arithmetic for every n -- the pair tables for n <= 2, the single-group MFMA tile for 3 <= n <= 6 (which the
    default engine only uses when the exact int8 route declines), no column peel"""
    o = Oracle()
    dg = mx.dgemm_compressed
    prob = make_problem(snps, indiv, n, seed=42 + snps)
    k = indiv if trans else snps
    m = snps if trans else indiv
    B = make_B(k, n, seed=43)
    ref = o.dgemm_dense(trans, prob, B, centered)[:, :m]
    prev = dg.set_engine("f64-strict")
    try:
        C = _run(mx, prob, trans, B, centered)
        # pair tables for n <= 2 where the stored (SNP-major) copy's rows are the output rows ('T'); 'N' of a one-copy object: the MFMA tile, transposed form
        assert dg.last_path() == ("k_lut" if n <= 2 and trans else "k_gemm")
    finally:
        dg.set_engine(prev)
    err = np.abs(C.T - ref).max() / np.abs(ref).max()
    assert err <= RTOL, err


def test_missing_codes_are_zero_then_centred(mx):
    o = Oracle()
    prob = make_problem(1203, 610, 8, seed=7, missing_frac=0.1)
    for trans in (0, 1):
        k = prob["indiv"] if trans else prob["snps"]
        m = prob["snps"] if trans else prob["indiv"]
        B = make_B(k, 8, seed=5)
        ref = o.dgemm_dense(trans, prob, B, 1)[:, :m]
        C = _run(mx, prob, trans, B, 1)
        assert np.abs(C.T - ref).max() / np.abs(ref).max() <= RTOL


@pytest.mark.parametrize("fixture,idx", [("dgemm_golden.npz", i) for i in range(7)] + [("dgemm_golden2.npz", i) for i in range(6)])
def test_dgemm_vs_reference_golden(mx, fixture, idx):
    """the HIP path against the outputs of the reference's own CPU library (tests/golden/dgemm_golden.npz; dgemm_golden2.npz: the operand
    patterns of the reference's Fortran tests, n = 40 / 65, SIMD variants 32 and 128 of the reference), through the raw
    C ABI with the fixtures' leading dimensions (padded ldb with poison, padded ldc zero-filled)"""
    import ctypes
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", fixture))
    name = str(g["names"][idx])
    snps, indiv, n, ldb_pad, ldc_pad = [int(x) for x in g[f"{name}/dims"]]
    L = mx.check_library_handle()
    dg = mx.dgemm_compressed
    plink, plink_t, f = (np.ascontiguousarray(g[f"{name}/{k}"]) for k in ("plink", "plink_t", "f"))
    for centered in (0, 1):
        dg.set_options(use_gpu=True, not_center=not centered, verbose=0)
        obj = dg.init_compressed(plink, plink_t, snps, indiv, f, n)
        for trans in (0, 1):
            k = indiv if trans else snps
            m = snps if trans else indiv
            B = np.ascontiguousarray(g[f"{name}/B{trans}"])
            ref = g[f"{name}/C{trans}{centered}"]
            C = np.full_like(ref, -777.0)
            L.dgemm_compressed(b"T" if trans else b"N", obj, n, B.ctypes.data_as(ctypes.c_void_p), k + ldb_pad, C.ctypes.data_as(ctypes.c_void_p), m + ldc_pad)
            assert L.mxa_last_error() == 0
            assert np.abs(C - ref).max() <= RTOL * np.abs(ref).max(), (name, trans, centered)
            assert np.all(C[:, m:] == 0.0)
        fq = np.zeros(snps)
        L.get_compressed_freq(obj, fq.ctypes.data_as(ctypes.c_void_p))
        assert np.array_equal(fq, f)
        dg.free_compressed(obj)
