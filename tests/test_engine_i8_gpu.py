"""Opt-in engine 'i8' (mxa_gemm_i8.hip: every column of B split exactly into 7 radix-256 digits, int8 matrix cores, exact
int32 accumulation, fp64 recombination) against the oracle, through the C ABI."""
import numpy as np
import pytest

from _util import Oracle, make_B, make_problem

pytestmark = pytest.mark.gpu

# B is represented to 2^-57 of each column's largest |entry|; all integer sums are exact.  Stated tolerance, relative to each
# result column's largest |entry| (the same figure as the fp64 engine's, tests/test_dgemm_gpu.py):
RTOL = 1e-11


@pytest.fixture(autouse=True)
def _both_copies():
    """the opt-in engines at wide n multiply 'N' on the plain int8 kernel, which reads the individual-major copy: the objects of this module store BOTH
    copies (MXA_SINGLE_ORIENTATION=0; a default object keeps the SNP-major copy alone and sends such an 'N' to the fp64 engine)"""
    import os
    os.environ["MXA_SINGLE_ORIENTATION"] = "0"
    yield
    os.environ.pop("MXA_SINGLE_ORIENTATION", None)


@pytest.fixture(scope="module")
def dg():
    import miraculix_amd as m
    m.load_shared_library()
    prev = m.dgemm_compressed.set_engine("i8")
    yield m.dgemm_compressed
    m.dgemm_compressed.set_engine(prev)


@pytest.mark.parametrize("snps,indiv,n", [(1000, 500, 1), (1003, 501, 3), (2047, 771, 4), (777, 1301, 10), (4100, 515, 15), (3001, 2050, 32), (1500, 700, 40), (20000, 300, 33)])
@pytest.mark.parametrize("centered", [0, 1])
def test_i8_engine_vs_oracle(dg, snps, indiv, n, centered):
    o = Oracle()
    prob = make_problem(snps, indiv, n, seed=42 + snps, missing_frac=0.02)
    dg.set_options(use_gpu=True, not_center=not centered, verbose=0)
    obj = dg.init_compressed(prob["plink"], prob["plink_t"], snps, indiv, prob["f"], n)
    try:
        for trans in (0, 1):
            k = indiv if trans else snps
            m = snps if trans else indiv
            B = make_B(k, n, seed=43)
            B[n // 2, :k] *= 1e-9  # column scales differ by 16 decades: per-column exponents
            B[0, :k] *= 3e7
            ref = o.dgemm_dense(trans, prob, B, centered)[:, :m]
            C = dg.dgemm_compressed_main(bool(trans), obj, np.asfortranarray(B[:, :k].T), snps, indiv)
            err = (np.abs(C.T - ref).max(axis=1) / np.abs(ref).max(axis=1)).max()  # worst column-wise relative error
            assert err <= RTOL, (trans, err)
    finally:
        dg.free_compressed(obj)


@pytest.mark.parametrize("snps,indiv,n", [(777, 1301, 10), (3001, 2050, 32), (1500, 700, 40), (20000, 300, 33), (2600, 900, 28), (2600, 900, 55)])
def test_i8_engine_on_one_copy_objects_multiplies_N_from_the_snp_major_copy(dg, snps, indiv, n):
    """default objects keep ONE packed copy: 'N' at wide n runs k_gemm_i8_tn in column chunks of at most six digit tiles (three passes of two tiles each);
    same digits and exact integer sums as the plain kernel, so the same tolerance -- and integer-valued B must come out bit-exact"""
    import os
    os.environ["MXA_SINGLE_ORIENTATION"] = "1"
    o = Oracle()
    prob = make_problem(snps, indiv, n, seed=42 + snps, missing_frac=0.02)
    for centered in (0, 1):
        dg.set_options(use_gpu=True, not_center=not centered, verbose=0)
        obj = dg.init_compressed(prob["plink"], prob["plink_t"], snps, indiv, prob["f"], n)
        try:
            assert dg.single_orientation(obj) == 1
            B = make_B(snps, n, seed=43)
            B[n // 2, :snps] *= 1e-9
            B[0, :snps] *= 3e7
            ref = o.dgemm_dense(0, prob, B, centered)[:, :indiv]
            Bcm = np.asfortranarray(B[:, :snps].T)
            C = dg.dgemm_compressed_main(False, obj, Bcm, snps, indiv)
            assert dg.last_path() == "k_gemm_i8"
            err = (np.abs(C.T - ref).max(axis=1) / np.abs(ref).max(axis=1)).max()
            assert err <= RTOL, err
            assert np.array_equal(C, dg.dgemm_compressed_main(False, obj, Bcm, snps, indiv))
            if not centered:
                Bi = np.random.default_rng(3).integers(-(2 ** 20), 2 ** 20, size=(n, snps)).astype(np.float64)
                Ci = dg.dgemm_compressed_main(False, obj, np.asfortranarray(Bi.T), snps, indiv)
                assert dg.last_path() == "k_gemm_i8"
                assert np.array_equal(Ci, (prob["Z"].astype(np.int64) @ Bi.T.astype(np.int64)).astype(np.float64))
        finally:
            dg.free_compressed(obj)


def test_i8_engine_zero_column_ld_padding_and_engine_switch(dg):
    o = Oracle()
    prob = make_problem(1203, 610, 5, seed=7)
    dg.set_options(use_gpu=True, not_center=True, verbose=0)
    obj = dg.init_compressed(prob["plink"], prob["plink_t"], 1203, 610, prob["f"], 5)
    try:
        B = make_B(1203, 5, seed=1, ldb=1210)   # poisoned ld padding must not be read
        B[2, :1203] = 0.0
        ref = o.dgemm_dense(0, prob, B, 0)[:, :610]
        Bcm = np.asfortranarray(B[:, :1203].T)
        C8 = dg.dgemm_compressed_main(False, obj, Bcm, 1203, 610)
        assert np.abs(C8[:, 2]).max() == 0.0
        assert np.abs(C8.T - ref).max() <= RTOL * np.abs(ref).max()
        assert dg.set_engine("f64") == "i8"
        C64 = dg.dgemm_compressed_main(False, obj, Bcm, 1203, 610)
        assert dg.set_engine("i8") == "f64"
        assert np.abs(C64 - C8).max() <= RTOL * np.abs(ref).max()
    finally:
        dg.free_compressed(obj)


def test_i8_engine_integer_B_is_exact(dg):
    """With integer-valued B (|b| < 2^20) the digits represent B exactly and every sum is an integer below 2^53: the result must
    equal the integer product bit for bit."""
    rng = np.random.default_rng(3)
    prob = make_problem(3000, 400, 6, seed=11)
    dg.set_options(use_gpu=True, not_center=True, verbose=0)
    obj = dg.init_compressed(prob["plink"], prob["plink_t"], 3000, 400, prob["f"], 6)
    try:
        B = rng.integers(-(2 ** 20), 2 ** 20, size=(6, 3000)).astype(np.float64)
        C = dg.dgemm_compressed_main(False, obj, np.asfortranarray(B.T), 3000, 400)
        ref = prob["Z"].astype(np.int64) @ B.T.astype(np.int64)
        assert np.array_equal(C, ref.astype(np.float64))
    finally:
        dg.free_compressed(obj)


def test_i8_engine_small_n_uses_the_free_digits(dg):
    """n = 1, 2: one tile of 32 expanded columns is the unit of work, so 32 / 16 radix-256 digits are used at no cost.  A column
    that mixes entries 1e+18 and 1e-18 (n = 2: 1e+8 and 1e-8) must then come out as accurately as a column of O(1) entries: rows whose genotypes are 0 at
    the large entries still get their (tiny) result to full relative precision."""
    o = Oracle()
    snps, indiv = 4001, 600
    prob = make_problem(snps, indiv, 1, seed=21)
    Z = prob["Z"]
    dg.set_options(use_gpu=True, not_center=True, verbose=0)
    obj = dg.init_compressed(prob["plink"], prob["plink_t"], snps, indiv, prob["f"], 2)
    try:
        rng = np.random.default_rng(8)
        for n in (1, 2):
            B = rng.standard_normal((n, snps))
            big = np.zeros(snps, bool); big[[0, 1500, 3000]] = True
            scale = 1e18 if n == 1 else 1e8      # 32 digits cover 36 decades with full mantissas, 16 digits cover 16
            B[:, big] *= scale
            B[:, ~big] /= scale
            C = dg.dgemm_compressed_main(False, obj, np.asfortranarray(B.T), snps, indiv)          # indiv x n
            # exact reference in two scales: the part from the big entries and the part from the small ones, each in long double
            Zl = Z.astype(np.longdouble)
            ref_big = (Zl[:, big] @ B[:, big].T.astype(np.longdouble))
            ref_small = (Zl[:, ~big] @ B[:, ~big].T.astype(np.longdouble))
            rows_without_big = np.abs(Z[:, big]).sum(axis=1) == 0
            assert rows_without_big.any()
            got = C.astype(np.longdouble)
            # rows that never touch a big entry: the result is ~1e-17 and must be right to ~1e-13 relative, not lost below 1e+18 * 2^-57
            sel = rows_without_big
            rel = np.abs(got[sel] - ref_small[sel]).max() / np.abs(ref_small[sel]).max()
            assert rel < 1e-13, (n, float(rel))
            tot = ref_big + ref_small
            assert np.abs(got - tot).max() <= 1e-13 * np.abs(tot).max()
    finally:
        dg.free_compressed(obj)


def test_i8_engine_negative_entries_far_below_the_column_maximum(dg):
    """Regression: digits are extracted in integer arithmetic.  (A floating-point extraction in two's-complement style rounded
    -1e-36 + 1 to 1.0 and overflowed the next digit for negative entries more than 2^53 below the column maximum.)"""
    o = Oracle()
    snps, indiv, n = 3000, 500, 8
    prob = make_problem(snps, indiv, n, seed=31)
    dg.set_options(use_gpu=True, not_center=True, verbose=0)
    obj = dg.init_compressed(prob["plink"], prob["plink_t"], snps, indiv, prob["f"], n)
    try:
        B = make_B(snps, n, seed=5)
        B[:, 10:] *= 1e-20            # every column: ten O(1) entries, the rest +-1e-20
        B[3, :] *= 1e25
        ref = o.dgemm_dense(0, prob, B, 0)[:, :indiv]
        C = dg.dgemm_compressed_main(False, obj, np.asfortranarray(B.T), snps, indiv)
        err = (np.abs(C.T - ref).max(axis=1) / np.abs(ref).max(axis=1)).max()
        assert err <= 1e-13, err
    finally:
        dg.free_compressed(obj)


def test_engine_switch_values(dg):
    """mxa_set_engine: 0 / 1 / 3 / 4 accepted; anything else -- the retired ids 2 (small-n-i8) and 5 (i8-guarded) included -- leaves the engine unchanged"""
    import miraculix_amd as m
    L = m.lib.check_library_handle()
    assert L.mxa_get_engine() == 1
    for bad in (7, -1, 2, 5):
        assert L.mxa_set_engine(bad) == 1 and L.mxa_get_engine() == 1
    assert dg.set_engine("f64") == "i8" and dg.set_engine("f64-strict") == "f64" and dg.set_engine("i8-exact") == "f64-strict"
    assert dg.set_engine("i8") == "i8-exact"
    with pytest.raises(ValueError):
        dg.set_engine("i8-guarded")
