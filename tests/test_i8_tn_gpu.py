"""Transposed-operand form of the exact int8 route for n <= 2 (round 4; MXA_I8_TN=1): the product computed from the copy whose ROWS are the K index --
'N' (the second product of a CG step) from the SNP-major copy -- by k_gemm_i8_tn: digits as the MFMA's A operand, the genotype operand gathered with
v_perm_b32 + in-place field masks (the A/B single-orientation storage needs for the CG path).  Same digits, same exactness
guard, exact integer sums: the results must agree with the plain form to rounding (the recombination adds the same exact integers in another fixed
order), be IDENTICAL for integer-valued B, obey the element-wise bound of the route and match the long-double oracle."""
import os

import numpy as np
import pytest

from _util import Oracle, make_B, make_problem

pytestmark = pytest.mark.gpu
U = 2.0 ** -53


def _two_copies(dg, prob, snps, indiv, n):
    """an object that stores BOTH packed copies (MXA_SINGLE_ORIENTATION=0: the opt-in since round 5), so that each product can be computed in either form"""
    os.environ["MXA_SINGLE_ORIENTATION"] = "0"
    try:
        return dg.init_compressed(prob["plink"], prob["plink_t"], snps, indiv, prob["f"], n)
    finally:
        os.environ.pop("MXA_SINGLE_ORIENTATION", None)


@pytest.fixture(scope="module")
def mx():
    import miraculix_amd as m
    m.load_shared_library()
    return m


def _product(dg, obj, prob, trans, B, tn):
    os.environ["MXA_I8_TN"] = "1" if tn else "0"
    try:
        C = dg.dgemm_compressed_main(bool(trans), obj, np.asfortranarray(B.T), prob["snps"], prob["indiv"])
        assert dg.last_path() == "k_gemm_i8"
        return C
    finally:
        os.environ.pop("MXA_I8_TN", None)


@pytest.mark.parametrize("snps,indiv", [(3001, 1037), (2050, 1301), (700, 3001), (5000, 600), (1300, 130)])
@pytest.mark.parametrize("n", [1, 2])
def test_transposed_int8_route_matches_plain_and_oracle(mx, snps, indiv, n):
    o = Oracle()
    prob = make_problem(snps, indiv, n, seed=snps + 3 * n, missing_frac=0.03)
    dg = mx.dgemm_compressed
    obj = _two_copies(dg, prob, snps, indiv, n)
    try:
        for centered in (0, 1):
            dg.set_options(use_gpu=True, not_center=not centered, verbose=0)
            for trans in (0, 1):
                k, m = (indiv, snps) if trans else (snps, indiv)
                B = make_B(k, n, seed=5 + centered + 2 * trans) * 10.0 ** np.random.default_rng(trans).uniform(-4, 0, size=(n, k))   # 13 binades inside a column
                C0 = _product(dg, obj, prob, trans, B, False)
                C1 = _product(dg, obj, prob, trans, B, True)
                ref = o.dgemm_dense(trans, prob, B, centered)[:, :m]
                assert np.abs(C1.T - ref).max() <= 1e-11 * np.abs(ref).max()
                abssum = o.dgemm_dense(trans, prob, np.abs(B), 0)[:, :m]
                tol = 3.02 * 31 * U * abssum + (8 * U * np.abs(ref - o.dgemm_dense(trans, prob, B, 0)[:, :m]) if centered else 0.0)
                assert np.all(np.abs(C1.T - ref) <= tol + 1e-300)                         # the element-wise bound of the exact route
                assert np.abs(C1 - C0).max() <= 64 * U * np.abs(abssum).max()
                Bi = np.round(B * 1e5)                                                     # integer-valued: every term is an exact integer
                if not centered:
                    assert np.array_equal(_product(dg, obj, prob, trans, Bi, True), _product(dg, obj, prob, trans, Bi, False))
                assert np.array_equal(C1, _product(dg, obj, prob, trans, B, True))        # bitwise repeatable
    finally:
        dg.free_compressed(obj)


def test_transposed_int8_route_guard_declines_like_the_plain_one(mx):
    """a column spanning more binades than the digits hold: the exactness guard raises its flag, k_gemm_i8_tn and its finish return at once and the
    fp64 pair-table fallback runs -- whatever the operand form"""
    o = Oracle()
    snps, indiv = 2000, 900
    prob = make_problem(snps, indiv, 1, seed=9)
    dg = mx.dgemm_compressed
    dg.set_options(use_gpu=True, not_center=False, verbose=0)
    obj = dg.init_compressed(prob["plink"], prob["plink_t"], snps, indiv, prob["f"], 1)
    try:
        B = make_B(snps, 1, seed=2)
        B[0, ::3] *= 1e-80                                       # 265 binades: beyond 8 * 32 - 55
        os.environ["MXA_I8_TN"] = "1"
        C = dg.dgemm_compressed_main(False, obj, np.asfortranarray(B.T), snps, indiv)
        assert dg.last_path() == "k_small_n_fp64"
        ref = o.dgemm_dense(0, prob, B, 1)[:, :indiv]
        assert np.abs(C.T - ref).max() <= 1e-11 * np.abs(ref).max()
    finally:
        os.environ.pop("MXA_I8_TN", None)
        dg.free_compressed(obj)
