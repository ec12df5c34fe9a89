"""n <= 2 (the CG / GBLUP iteration) under the default engine: the exact int8 slicing is taken only when a per-call device check
proves that it represents B without error; then the result must obey the bound stated in include/miraculix_amd.h --
|error| <= 3.02 * 31 * 2^-53 * sum_k |z_k b_k| per output, which is below the K * 2^-53 * sum_k |z_k b_k| of an fp64 FMA chain -- on
adversarial inputs (columns spanning many decades, genotype rows that are zero exactly where B is large).  When the check fails
(span too wide, non-finite entries, values near the underflow threshold, K < 128) the fp64 pair-table kernel must run and keep
the element-wise fp64 bound.  Oracle: long-double dense products (oracle/oracle.c)."""
import numpy as np
import pytest

from _util import Oracle, make_problem

pytestmark = pytest.mark.gpu
U = 2.0 ** -53


@pytest.fixture(scope="module")
def mx():
    import miraculix_amd as m
    m.load_shared_library()
    return m


def _adversarial_problem(snps, indiv, seed):
    """genotypes whose first 40 individuals are 0 at every 'big' SNP (every 7th) and whose first 40 SNPs are 0 for every 'big'
    individual (every 5th): those outputs see only the small entries of B"""
    prob = make_problem(snps, indiv, 1, seed=seed)
    Z = prob["Z"].copy()                      # indiv x snps
    Z[:40, ::7] = 0
    Z[::5, :40] = 0
    from _util import pack_plink
    prob["Z"] = Z
    prob["plink"] = np.ascontiguousarray(pack_plink(Z.T.copy()))
    prob["plink_t"] = np.ascontiguousarray(pack_plink(Z))
    prob["f"] = Z.astype(np.float64).mean(axis=0) / 2.0
    return prob


def _wide_B(k, n, decades, seed, big_every):
    rng = np.random.default_rng(seed)
    B = rng.standard_normal((n, k)) * 10.0 ** rng.uniform(-decades, -decades / 2, size=(n, k))
    B[:, ::big_every] = rng.standard_normal((n, len(range(0, k, big_every)))) * 10.0 ** rng.uniform(-1, 0, size=(n, len(range(0, k, big_every))))
    return B


def _run(mx, obj, prob, trans, B):
    dg = mx.dgemm_compressed
    return dg.dgemm_compressed_main(bool(trans), obj, np.asfortranarray(B.T), prob["snps"], prob["indiv"]).T   # n x m


@pytest.mark.parametrize("n,decades", [(1, 50), (2, 18)])
@pytest.mark.parametrize("trans", [0, 1])
def test_exact_int8_route_obeys_the_stated_bound(mx, n, decades, trans):
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
        ref = o.dgemm_dense(trans, prob, B, 0)[:, :m]
        abssum = o.dgemm_dense(trans, prob, np.abs(B), 0)[:, :m]           # sum_k |z_k b_k| per output
        err = np.abs(C - ref)
        bound = 3.02 * 31 * U * abssum
        chain = k * U * abssum
        assert np.all(err <= bound + 1e-300), float((err / np.maximum(bound, 1e-300)).max())
        assert np.all(bound <= chain)
        # the adversarial outputs really are many decades below the typical ones, and still accurate to fp64 relative precision
        small = abssum < 1e-6 * abssum.max()
        assert small.any()
        assert np.all(err[small] <= 1e-13 * abssum[small])
        # centred: same route, stated path tolerance
        dg.set_options(use_gpu=True, not_center=False, verbose=0)
        Cc = _run(mx, obj, prob, trans, B)
        assert dg.last_path() == "k_gemm_i8"
        refc = o.dgemm_dense(trans, prob, B, 1)[:, :m]
        assert np.abs(Cc - refc).max() <= 1e-11 * np.abs(refc).max()
        # bitwise repeatable
        assert np.array_equal(Cc, _run(mx, obj, prob, trans, B))
    finally:
        dg.free_compressed(obj)


@pytest.mark.parametrize("case", ["span", "inf", "nan", "tiny", "strict"])
@pytest.mark.parametrize("n", [1, 2])
def test_guard_declines_and_fp64_tables_take_over(mx, case, n):
    o = Oracle()
    snps, indiv = 2050, 777
    prob = _adversarial_problem(snps, indiv, seed=5)
    dg = mx.dgemm_compressed
    dg.set_options(use_gpu=True, not_center=True, verbose=0)
    obj = dg.init_compressed(prob["plink"], prob["plink_t"], snps, indiv, prob["f"], n)
    prev = None
    try:
        for trans in (0, 1):
            k = indiv if trans else snps
            m = snps if trans else indiv
            B = _wide_B(k, n, 30 if n == 2 else 70, seed=9, big_every=5 if trans else 7)   # 100 / 230 bits of span: beyond 73 / 201
            if case == "inf":
                B = np.random.default_rng(1).standard_normal((n, k)); B[n - 1, 17] = np.inf
            elif case == "nan":
                B = np.random.default_rng(1).standard_normal((n, k)); B[0, 3] = np.nan
            elif case == "tiny":
                B = np.random.default_rng(1).standard_normal((n, k)) * 1e-290     # recombination would leave the normal range
            elif case == "strict":
                B = np.random.default_rng(1).standard_normal((n, k))
                prev = dg.set_engine("f64-strict")
            C = _run(mx, obj, prob, trans, B)
            # engine f64-strict: pair tables ('T'; 'N' of a one-copy object: the MFMA tile in its transposed form); else the fp64 chains behind the declined guard
            assert dg.last_path() == (("k_lut" if trans else "k_gemm") if case == "strict" else "k_small_n_fp64"), (case, trans)
            with np.errstate(invalid="ignore", over="ignore"):
                ref = o.dgemm_dense(trans, prob, B, 0)[:, :m]
                abssum = o.dgemm_dense(trans, prob, np.abs(B), 0)[:, :m]
            if case in ("inf", "nan"):
                bad = ~np.isfinite(ref)
                assert np.array_equal(~np.isfinite(C), bad)
                good = ~bad
                assert np.all(np.abs(C[good] - ref[good]) <= k * U * abssum[good])
            else:
                assert np.all(np.abs(C - ref) <= k * U * abssum + 1e-320)       # element-wise fp64 chain bound, adversarial rows included
            if prev is not None:
                dg.set_engine(prev); prev = None
    finally:
        if prev is not None:
            dg.set_engine(prev)
        dg.free_compressed(obj)


def test_zero_vector_short_k_and_mixed_columns(mx):
    o = Oracle()
    dg = mx.dgemm_compressed
    dg.set_options(use_gpu=True, not_center=True, verbose=0)
    # K < 128: always the fp64 tables
    prob = make_problem(100, 333, 1, seed=2)
    obj = dg.init_compressed(prob["plink"], prob["plink_t"], 100, 333, prob["f"], 2)
    try:
        B = np.random.default_rng(0).standard_normal((1, 100))
        C = _run(mx, obj, prob, 0, B)
        assert dg.last_path() == "k_gemm"                                # K < 128, 'N' of a one-copy object: the MFMA tile, transposed form ('T': the pair tables)
        ref = o.dgemm_dense(0, prob, B, 0)[:, :333]
        assert np.abs(C - ref).max() <= 1e-11 * np.abs(ref).max()
    finally:
        dg.free_compressed(obj)
    prob = make_problem(1500, 640, 2, seed=3)
    obj = dg.init_compressed(prob["plink"], prob["plink_t"], 1500, 640, prob["f"], 2)
    try:
        # an all-zero column next to an ordinary one: exact route, exact zeros
        B = np.zeros((2, 1500)); B[1] = np.random.default_rng(4).standard_normal(1500)
        C = _run(mx, obj, prob, 0, B)
        assert dg.last_path() == "k_gemm_i8"
        assert np.all(C[0] == 0.0)
        ref = o.dgemm_dense(0, prob, B, 0)[:, :640]
        assert np.abs(C - ref).max() <= 1e-11 * np.abs(ref).max()
        # integer-valued B: every route gives the exact integers
        Bi = np.random.default_rng(5).integers(-1000, 1000, size=(2, 1500)).astype(np.float64)
        Ci = _run(mx, obj, prob, 0, Bi)
        assert np.array_equal(Ci, (prob["Z"].astype(np.float64) @ Bi.T).T)
        # one column fine, the other beyond the span: the whole call falls back
        B[0] = 1e-40; B[0, 0] = 1.0
        C = _run(mx, obj, prob, 0, B)
        assert dg.last_path() == "k_small_n_fp64"
        ref = o.dgemm_dense(0, prob, B, 0)[:, :640]
        assert np.abs(C - ref).max() <= 1e-11 * np.abs(ref).max()
    finally:
        dg.free_compressed(obj)


def _last_n(mx):
    import ctypes
    L = mx.check_library_handle()
    gm, gk, gn, gs, ga, gc = ctypes.c_long(), ctypes.c_long(), ctypes.c_int(), ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    L.mxa_last_geometry(ctypes.byref(gm), ctypes.byref(gk), ctypes.byref(gn), ctypes.byref(gs), ctypes.byref(ga), ctypes.byref(gc))
    return gn.value, ga.value, gc.value


@pytest.mark.parametrize("n", [5, 6, 7, 10, 15, 33])
def test_column_peel_exact_route_and_fallback(mx, n):
    """n = 4q + 1 / 4q + 2, n > 6: the odd columns take the guarded exact int8 route, the MFMA tile multiplies 4q columns without padding (the
    reference harness's n = 10 becomes 8 + 2); peeled columns whose entries span too many binades are multiplied by the fp64 kernel behind the guard
    (k_small_n_fp64; the verdict is formed on the device, so nobody waits for it).
    n <= 6: all columns take the exact int8 route when it is exact (two classes of digit counts), the fp64 kernel otherwise.
    n = 4q + 3 > 6: the three odd columns take the same route"""
    o = Oracle()
    snps, indiv = 2050, 777
    prob = _adversarial_problem(snps, indiv, seed=5)
    dg = mx.dgemm_compressed
    obj = dg.init_compressed(prob["plink"], prob["plink_t"], snps, indiv, prob["f"], n)
    try:
        for centered in (0, 1):
            dg.set_options(use_gpu=True, not_center=not centered, verbose=0)
            for trans in (0, 1):
                k = indiv if trans else snps
                m = snps if trans else indiv
                B = np.random.default_rng(n + trans).standard_normal((n, k))
                C = _run(mx, obj, prob, trans, B)
                if n <= 6:                                               # round 3: the whole product takes the exact int8 route
                    assert dg.last_path() == "k_gemm_i8" and _last_n(mx)[0] == n
                else:
                    assert _last_n(mx)[0] == n - n % 4                   # the MFMA launch saw the multiple of 4 only
                ref = o.dgemm_dense(trans, prob, B, centered)[:, :m]
                assert np.abs(C - ref).max() <= 1e-11 * np.abs(ref).max()
                B[n - 1] = _wide_B(k, 1, 70, seed=2, big_every=5)[0]     # 230 binades in the last column: guard declines
                C = _run(mx, obj, prob, trans, B)
                if n <= 6:                                               # round 5: the verdict is formed on the device; the fp64 chains behind the int8 route did the product
                    assert dg.last_path() == "k_small_n_fp64" and _last_n(mx)[0] == n
                else:                                                    # ... or the peeled columns; the MFMA launch still saw the multiple of 4
                    assert _last_n(mx)[0] == n - n % 4
                ref = o.dgemm_dense(trans, prob, B, centered)[:, :m]
                err = np.abs(C - ref)
                abssum = o.dgemm_dense(trans, prob, np.abs(B), 0)[:, :m]
                tol = k * U * abssum + (8 * U * np.abs(ref - o.dgemm_dense(trans, prob, B, 0)[:, :m]) if centered else 0.0)
                assert np.all(err <= tol + 1e-300)
    finally:
        dg.free_compressed(obj)


@pytest.mark.parametrize("n", [1, 2])
@pytest.mark.parametrize("trans", [0, 1])
@pytest.mark.parametrize("centered", [False, True])
def test_result_does_not_depend_on_ldc_or_operand_residence(mx, n, trans, centered):
    """single-split products finish inside k_gemm_i8, multi-split ones through P + k_finish_i8_small; which of the two runs must
    depend on the shape only: the same product into a padded device C (ldc = m + 7; for n > 1 the padding rows come back zero like on the reference's CPU path) and from host
    operands is bit-identical"""
    import torch
    dg = mx.dgemm_compressed
    dg.set_options(use_gpu=True, not_center=not centered, verbose=0)
    snps, indiv = 2050, 1301
    prob = make_problem(snps, indiv, n, seed=11)
    obj = dg.init_compressed(prob["plink"], prob["plink_t"], snps, indiv, prob["f"], 2)
    try:
        k, m = (indiv, snps) if trans else (snps, indiv)
        B = np.asfortranarray(np.random.default_rng(12).standard_normal((k, n)))
        C_host = dg.dgemm_compressed_main(bool(trans), obj, B, snps, indiv)
        assert dg.last_path() == "k_gemm_i8"
        Bd = torch.from_numpy(np.ascontiguousarray(B.T)).cuda().t()               # column-major on the device
        buf = torch.full((n, m + 7), 7.0, dtype=torch.float64, device="cuda")
        Cd = dg.dgemm_compressed_main(bool(trans), obj, Bd, snps, indiv, out=buf.t()[:m])
        torch.cuda.synchronize()
        assert dg.last_path() == "k_gemm_i8"
        assert np.array_equal(Cd.cpu().numpy(), C_host)
        if n > 1:
            assert np.all(buf[:, m:].cpu().numpy() == 0.0)
    finally:
        dg.free_compressed(obj)
