"""CPU tests of the oracle (test infrastructure): pinned against the golden fixtures emitted by the reference's own CPU
library (tests/golden/make_golden.py), against the analytic oracles of the reference's tests, and - where the reference
build travelled (oracle/_ref) - against the reference library itself."""
import os

import numpy as np
import pytest

from _util import Oracle, have_reference, make_B, make_problem, pack_plink, run_reference

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "dgemm_golden.npz")


@pytest.fixture(scope="module")
def golden():
    return np.load(GOLDEN)


@pytest.fixture(scope="module")
def oracle():
    return Oracle()


def _case(golden, name):
    snps, indiv, n, ldb_pad, ldc_pad = [int(x) for x in golden[f"{name}/dims"]]
    prob = dict(snps=snps, indiv=indiv, plink=np.ascontiguousarray(golden[f"{name}/plink"]), plink_t=np.ascontiguousarray(golden[f"{name}/plink_t"]), f=np.ascontiguousarray(golden[f"{name}/f"]))
    return prob, n, ldb_pad, ldc_pad


def test_golden_names(golden):
    assert len(golden["names"]) >= 7


@pytest.mark.parametrize("idx", range(7))
def test_five_codes_port_bit_exact_vs_reference_golden(golden, oracle, idx):
    """the from-scratch 5codes engine reproduces the reference library bit for bit (same `cores` slicing parameter)"""
    name = str(golden["names"][idx])
    prob, n, ldb_pad, ldc_pad = _case(golden, name)
    cores = int(golden["cores"][0])
    h = oracle.five_create(prob, cores)
    for trans in (0, 1):
        m = prob["snps"] if trans else prob["indiv"]
        B = np.ascontiguousarray(golden[f"{name}/B{trans}"])
        for centered in (0, 1):
            ref = golden[f"{name}/C{trans}{centered}"]
            C = oracle.five_dgemm(h, trans, prob, B, centered, ldc=m + ldc_pad)
            assert np.array_equal(C, ref), (name, trans, centered, np.abs(C - ref).max())
    oracle.five_free(h)


@pytest.mark.parametrize("idx", range(7))
def test_dense_and_gpuorder_oracles_vs_reference_golden(golden, oracle, idx):
    name = str(golden["names"][idx])
    prob, n, ldb_pad, ldc_pad = _case(golden, name)
    for trans in (0, 1):
        m = prob["snps"] if trans else prob["indiv"]
        B = np.ascontiguousarray(golden[f"{name}/B{trans}"])
        for centered in (0, 1):
            ref = golden[f"{name}/C{trans}{centered}"]
            scale = np.abs(ref).max()
            Cd = oracle.dgemm_dense(trans, prob, B, centered, ldc=m + ldc_pad)
            assert np.abs(Cd - ref).max() <= 1e-13 * scale
            assert np.all(Cd[:, m:] == 0.0) and np.all(ref[:, m:] == 0.0)   # ld padding zero-filled by both
            Cg = oracle.dgemm_gpuorder(trans, prob, B, centered, ldc=m + ldc_pad)
            assert np.abs(Cg[:, :m] - ref[:, :m]).max() <= 1e-12 * scale


GOLDEN2 = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "dgemm_golden2.npz")


@pytest.mark.parametrize("idx", range(6))
def test_oracles_vs_second_reference_golden_set(oracle, idx):
    """tests/golden/dgemm_golden2.npz (make_golden2.py): the Fortran tests' operand patterns, n = 40 / 65, the reference's SIMD variants 32 and 128.
    The 5codes port reproduces the AVX2 variant (256) bit for bit; the other variants add in another order and agree to rounding; the dense
    long-double oracle -- the checker of the -m gpu tests -- agrees with every one of them to 1e-13."""
    g2 = np.load(GOLDEN2)
    name = str(g2["names"][idx])
    prob, n, _, _ = _case(g2, name)
    h = oracle.five_create(prob, int(g2["cores"][0]) if "cores" in g2 else 8)
    for trans in (0, 1):
        B = np.ascontiguousarray(g2[f"{name}/B{trans}"])
        for centered in (0, 1):
            ref = g2[f"{name}/C{trans}{centered}"]
            scale = np.abs(ref).max()
            C5 = oracle.five_dgemm(h, trans, prob, B, centered)
            if "variant" not in name:
                assert np.array_equal(C5, ref), (name, trans, centered, np.abs(C5 - ref).max())
            else:
                assert np.abs(C5 - ref).max() <= 1e-13 * scale
            assert np.abs(oracle.dgemm_dense(trans, prob, B, centered) - ref).max() <= 1e-13 * scale
    oracle.five_free(h)


def test_missing_is_zero_then_centred(golden, oracle):
    """code 01 decodes to 0 and is centred like any other entry (both reference engines)"""
    name = "missing_1203x610_n6"
    prob, n, _, _ = _case(golden, name)
    codes = np.stack([(prob["plink_t"] >> (2 * q)) & 3 for q in range(4)], axis=2).reshape(prob["indiv"], -1)[:, : prob["snps"]]
    assert (codes == 1).mean() > 0.05
    Z = np.maximum(codes.astype(np.float64) - 1, 0)
    B = golden[f"{name}/B0"]
    ref = golden[f"{name}/C01"]
    dense = ((Z - 2 * prob["f"][None, :]) @ B[:, : prob["snps"]].T).T
    assert np.abs(dense - ref).max() <= 1e-12 * np.abs(ref).max()


def test_analytic_oracle_of_reference_tests(oracle):
    """tests/dgemm_compressed/test.jl:97-104: (G .- 2f) * B with f = column means / 2"""
    prob = make_problem(1111, 407, 10, seed=3)
    for trans in (0, 1):
        k = 407 if trans else 1111
        B = make_B(k, 10, seed=11)
        Zc = prob["Z"].astype(np.float64) - 2 * prob["f"][None, :]
        dense = (Zc.T @ B.T).T if trans else (Zc @ B.T).T
        C = oracle.dgemm_dense(trans, prob, B, 1)
        assert np.abs(C - dense).max() <= 1e-12 * np.abs(dense).max()


def test_transpose_and_freq_helpers(oracle):
    for snps, indiv in [(13, 7), (1000, 500), (1003, 501)]:
        prob = make_problem(snps, indiv, 1, seed=snps)
        T = oracle.transpose_2bit(prob["plink"], snps, indiv)
        assert np.array_equal(T, prob["plink_t"])
        assert np.array_equal(oracle.transpose_2bit(T, indiv, snps), prob["plink"])
        f = oracle.allele_freq(prob["plink"], snps, indiv)
        assert np.abs(f - prob["f"]).max() < 1e-15


@pytest.mark.parametrize("n_snps,n_indiv", [(953, 752), (10251, 75), (131, 17)])
def test_crossprod_oracle_vs_dense_gemm(oracle, n_snps, n_indiv):
    """tests/crossproduct/test_grm.jl:127-129,152-154: exact integers vs BLAS gemm on the decoded 0/1/2 matrix"""
    prob = make_problem(n_snps, n_indiv, 1, seed=9)
    M = oracle.crossprod_i32(prob["plink_t"], n_snps, True)
    Z = prob["Z"].astype(np.int64)
    assert np.array_equal(M, (Z @ Z.T).astype(np.int32))
    assert np.array_equal(M, M.T)


def test_crossprod_oracle_missing_byte_table(oracle):
    """snp_multiply_cuda.h:202: a byte holding a missing pair becomes 0xFF (four 3s)"""
    Z = np.array([[0, 1, 2, 0, 1, 1, 2, 2]], dtype=np.int8)
    miss = np.zeros_like(Z, dtype=bool)
    miss[0, 5] = True
    X = pack_plink(Z, miss)
    M = oracle.crossprod_i32(X, 8, True)
    # first byte decodes 0,1,2,0 ; second byte holds a missing -> 3,3,3,3
    assert M[0, 0] == 0 + 1 + 4 + 0 + 9 * 4


@pytest.mark.skipif(not have_reference(), reason="reference CPU library (oracle/_ref) not built here")
def test_port_vs_live_reference_library(oracle):
    prob = make_problem(1500, 700, 9, seed=77)
    h = oracle.five_create(prob, 8)
    for trans in (0, 1):
        k = 700 if trans else 1500
        B = make_B(k, 9, seed=5)
        for centered in (0, 1):
            ref, _ = run_reference(prob, trans, B, centered, cores=8)
            assert np.array_equal(oracle.five_dgemm(h, trans, prob, B, centered), ref)
    oracle.five_free(h)
