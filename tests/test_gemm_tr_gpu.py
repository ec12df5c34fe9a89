"""Transposed-operand instantiation of k_gemm (round 4): a product computed from the copy whose ROWS are the K index -- 'N' from the SNP-major copy,
'T' from the individual-major one; output rows = columns of the packed matrix.  With the plain form (output rows = packed rows) it lets ONE packed copy
serve both products; since round 5 both forms share the permuted K order (mxa_kernels.hip: gemm_k_index) and run at the same rate.  MXA_GEMM_TR=0 / 1 force the plain /
the transposed form.  Same launch plan, same K order, same partial sums: the results must be BIT-IDENTICAL, for every
tile shape (n = 8 ... 128: C = 2 ... 8, A = 16 / 8; 33 and 10: peeled columns beside it), ragged sizes (individuals not a multiple of the 128-wide
slab nor of the 256-individual block of the A = 16 tiles: the clamped last slab), centred and not, padded leading dimensions -- and agree with the
long-double oracle."""
import os

import numpy as np
import pytest

from _util import Oracle, make_B, make_problem

pytestmark = pytest.mark.gpu


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


def _product(dg, obj, prob, trans, B, tr):
    os.environ["MXA_GEMM_TR"] = "1" if tr else "0"
    try:
        return dg.dgemm_compressed_main(bool(trans), obj, np.asfortranarray(B.T), prob["snps"], prob["indiv"])
    finally:
        os.environ.pop("MXA_GEMM_TR", None)


@pytest.mark.parametrize("snps,indiv", [(2051, 777), (1003, 130), (4100, 1290), (700, 3001)])
@pytest.mark.parametrize("n", [8, 10, 12, 20, 32, 33, 64, 128])
def test_transposed_operand_is_bit_identical(mx, snps, indiv, n):
    o = Oracle()
    prob = make_problem(snps, indiv, n, seed=snps + n, missing_frac=0.05)
    dg = mx.dgemm_compressed
    obj = _two_copies(dg, prob, snps, indiv, n)
    try:
        for centered in (0, 1):
            dg.set_options(use_gpu=True, not_center=not centered, verbose=0)
            for trans in (0, 1):
                k, m = (indiv, snps) if trans else (snps, indiv)
                B = make_B(k, n, seed=7 + centered + 2 * trans)
                C0 = _product(dg, obj, prob, trans, B, False)
                assert dg.last_path() == "k_gemm"
                C1 = _product(dg, obj, prob, trans, B, True)                   # field masked in place; the plain form has the scale on B's rows, this one on the output rows: exact powers of two
                assert dg.last_path() == "k_gemm"
                assert np.array_equal(C0, C1)
                assert np.array_equal(C0, dg.dgemm_compressed_main(bool(trans), obj, np.asfortranarray(B.T), snps, indiv))   # the default picks one of the two
                ref = o.dgemm_dense(trans, prob, B, centered)[:, :m]
                assert np.abs(C1.T - ref).max() <= 1e-11 * np.abs(ref).max()
    finally:
        dg.free_compressed(obj)


def test_transposed_operand_range_fallback_and_ld_padding(mx):
    """a column of B spanning more binades than the denormal-operand mode carries: the plain-operand chain (MODE 0) runs transposed too; padded Ldb / Ldc"""
    import ctypes
    o = Oracle()
    snps, indiv, n = 1500, 515, 8
    prob = make_problem(snps, indiv, n, seed=3)
    dg = mx.dgemm_compressed
    L = mx.check_library_handle()
    dg.set_options(use_gpu=True, not_center=False, verbose=0)
    obj = _two_copies(dg, prob, snps, indiv, n)
    try:
        ldb, ldc = snps + 3, indiv + 5
        B = make_B(snps, n, seed=4, ldb=ldb)
        B[2, :snps:3] *= 1e-300
        B[2, 1:snps:3] *= 1e+250                                   # > 800 binades inside column 2: range guard raises its flag
        out = {}
        for tr in (0, 1):
            os.environ["MXA_GEMM_TR"] = str(tr)
            C = np.full((n, ldc), -777.0)
            L.dgemm_compressed(b"N", obj, n, B.ctypes.data_as(ctypes.c_void_p), ldb, C.ctypes.data_as(ctypes.c_void_p), ldc)
            assert L.mxa_last_range_fallback(obj) == 1
            out[tr] = C
        os.environ.pop("MXA_GEMM_TR", None)
        assert np.array_equal(out[0], out[1]) and np.all(out[1][:, indiv:] == 0.0)
        ref = o.dgemm_dense(0, prob, B, 1, ldc=ldc)
        err = np.abs(out[1][:, :indiv] - ref[:, :indiv])
        abssum = o.dgemm_dense(0, prob, np.abs(B), 0, ldc=ldc)[:, :indiv]
        assert np.all(err <= snps * 2.0 ** -52 * abssum + 1e-300)
    finally:
        os.environ.pop("MXA_GEMM_TR", None)
        dg.free_compressed(obj)
