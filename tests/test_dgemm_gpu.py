"""GPU parity tests proper: the HIP path, called through the C ABI, against the oracle on the same seeded inputs.
Mirrors tests/dgemm_compressed/test.jl:88-104 and test_5codesapi{,_t}.f90 of the reference (random B with n=10/15,
dense (G - 2f) B oracle, max-abs comparison) with a much tighter, stated tolerance."""
import numpy as np
import pytest

from _util import Oracle, elementwise_bound, make_B, make_problem, pack_plink

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
    # ... and the hard bound per ELEMENT: |C_ij - ref_ij| <= 4 K 2^-53 sum_k |z_ik| |b_kj| (centred: + the rank-1 term's magnitude)
    assert np.all(np.abs(C.T - ref) <= elementwise_bound(o, trans, prob, B, centered))


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
    assert np.all(np.abs(C.T - ref) <= elementwise_bound(o, trans, prob, B, centered))


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
    o = Oracle()
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
            # element-wise: against the long-double oracle on the fixture's inputs with the bound of ONE chain (4 K u sum|z||b|), and against the reference
            # library's own outputs with room for both chains' roundings (5 K u sum|z||b|)
            prob = dict(snps=snps, indiv=indiv, plink=plink, plink_t=plink_t, f=f)
            Bd = np.ascontiguousarray(B[:, :k])
            dense = o.dgemm_dense(trans, prob, Bd, centered)[:, :m]
            bound = elementwise_bound(o, trans, prob, Bd, centered)
            assert np.all(np.abs(C[:, :m] - dense) <= bound), (name, trans, centered)
            assert np.all(np.abs(C[:, :m] - ref[:, :m]) <= 1.25 * bound), (name, trans, centered)
        fq = np.zeros(snps)
        L.get_compressed_freq(obj, fq.ctypes.data_as(ctypes.c_void_p))
        assert np.array_equal(fq, f)
        dg.free_compressed(obj)


@pytest.mark.parametrize("trans", [0, 1])
@pytest.mark.parametrize("centered", [0, 1])
@pytest.mark.parametrize("n", [8, 32])
def test_fp64_mfma_path_keeps_small_rows_of_a_700_binade_column(mx, trans, centered, n):
    """Adversarial input for the denormal-operand form of k_gemm (columns of B scaled to just below 2^900, genotype operand z * 2^-1074; range guard at 800
    binades): every column of B holds a few entries near 2^+350 and all the others near 2^-350 -- a span of 700 binades, inside the guard -- and the genotypes
    are arranged so that some output rows never meet a large entry: their whole result is carried by the small entries, 700 binades below max|C|.  The
    norm-wise tolerance cannot see those rows (anything is within 1e-11 of max|C|); the element-wise bound must hold for them, uncentred EXACTLY relative to
    their own magnitude.  Path asserted: k_gemm without the range fallback."""
    o = Oracle()
    dg = mx.dgemm_compressed
    L = mx.check_library_handle()
    snps, indiv = 3000, 1100
    rng = np.random.default_rng(17 + n)
    prob = make_problem(snps, indiv, n, seed=91)
    Z = prob["Z"].copy()                                            # indiv x snps
    k, m = (indiv, snps) if trans else (snps, indiv)
    hot = np.sort(rng.choice(k, 12, replace=False))                 # K positions that carry the large entries
    quiet = np.sort(rng.choice(m, 40, replace=False))               # output rows that never meet them: genotype 0 at every hot position
    if trans:
        Z[np.ix_(hot, quiet)] = 0                                   # 'T': output rows = SNPs, K = individuals
    else:
        Z[np.ix_(quiet, hot)] = 0                                   # 'N': output rows = individuals, K = SNPs
    prob = dict(prob, Z=Z, plink=np.ascontiguousarray(pack_plink(Z.T.copy())), plink_t=np.ascontiguousarray(pack_plink(Z)), f=Z.astype(np.float64).mean(axis=0) / 2.0)
    B = make_B(k, n, seed=5) * 2.0 ** -350
    B[:, hot] *= 2.0 ** 700
    ref = o.dgemm_dense(trans, prob, B, centered)[:, :m]
    C = _run(mx, prob, trans, B, centered)
    assert dg.last_path() == "k_gemm"
    bound = elementwise_bound(o, trans, prob, B, centered)
    assert np.all(np.isfinite(C)) and np.all(np.abs(C.T - ref) <= bound)
    if not centered:                                                # the quiet rows live 700 binades below the others and are still right to their own scale
        q = np.abs(ref[:, quiet])
        assert q.max() < 2.0 ** -300 and np.abs(ref).max() > 2.0 ** 340
        assert np.all(np.abs(C.T[:, quiet] - ref[:, quiet]) <= 4.0 * k * 2.0 ** -53 * q.max())
        assert np.all(np.abs(C.T[:, quiet] - ref[:, quiet]) <= bound[:, quiet])
