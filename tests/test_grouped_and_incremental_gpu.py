"""Round 5: what BASELINE config 4 at its full 5M x 200k x 128 needs on ONE device, checked here at sizes the oracle finishes in seconds.

(1) GROUPED K splits (mxa_api.cpp: partial_budget, gemm_device): a product whose split-K partial sums would not fit beside the packed matrix runs
    its K splits in groups with the running sum kept in C.  Same pieces, same ascending order of additions: bit-identical to the one-pass product.
    MXA_P_BUDGET_MB (read per call) forces the grouped path on small products.
(2) Incremental staging (mxa_plink2compressed_begin / _rows / _end): the object is allocated first and filled by SNP-row blocks (host or device
    memory, any order); results, frequencies and the .bed reader's streaming path equal the one-pointer single-orientation object bit for bit.
Reference roles: plink2gpu (src/cuda/dgemm_compressed_cuda.cu:43-170) stages one whole matrix and gives up when matrix + object exceed the
device (:93-100); dgemm_compressed_gpu (:218-489) allocates its workspace per call."""
import ctypes
import os

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


class _env:
    def __init__(self, **kv):
        self.kv, self.old = kv, {}

    def __enter__(self):
        for k, v in self.kv.items():
            self.old[k] = os.environ.get(k)
            os.environ[k] = str(v)

    def __exit__(self, *a):
        for k, v in self.old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


@pytest.mark.parametrize("single", [False, True])
@pytest.mark.parametrize("snps,indiv,n", [(40_000, 3_000, 32), (3_000, 40_000, 32), (30_011, 2_051, 20), (20_000, 5_000, 128)])
def test_grouped_splits_bit_identical(mx, single, snps, indiv, n):
    """K long enough for several splits; budgets of 1 MB (one split per group where a split's partials exceed it) and of a few splits per group"""
    o = Oracle()
    prob = make_problem(snps, indiv, n, seed=snps % 1000 + n, missing_frac=0.02)
    dg = mx.dgemm_compressed
    L = mx.check_library_handle()
    with _env(MXA_SINGLE_ORIENTATION=1 if single else 0):
        obj = dg.init_compressed(prob["plink"], None if single else prob["plink_t"], snps, indiv, prob["f"], n)
    most = 0
    try:
        for centered in (1, 0):
            dg.set_options(use_gpu=True, not_center=not centered, verbose=0)
            for trans in (0, 1):
                k, m = (indiv, snps) if trans else (snps, indiv)
                B = make_B(k, n, seed=5 + trans)
                Bf = np.asfortranarray(B.T)
                C0 = dg.dgemm_compressed_main(bool(trans), obj, Bf, snps, indiv)
                m_, k_, n_, splits = ctypes.c_long(), ctypes.c_long(), ctypes.c_int(), ctypes.c_int()
                L.mxa_last_geometry(ctypes.byref(m_), ctypes.byref(k_), ctypes.byref(n_), ctypes.byref(splits), None, None)
                ref = o.dgemm_dense(trans, prob, B, centered)[:, :m]
                assert np.abs(C0.T - ref).max() <= RTOL * np.abs(ref).max()
                one_split_mb = 8.0 * n * (m + 512) / 2 ** 20
                for mb in (1, int(2.5 * one_split_mb) + 1):
                    with _env(MXA_P_BUDGET_MB=mb):
                        C1 = dg.dgemm_compressed_main(bool(trans), obj, Bf, snps, indiv)
                    assert np.array_equal(C0, C1), (trans, centered, mb, splits.value)
                most = max(most, splits.value)
        assert most >= 3                              # the product along the long dimension has several K splits to group
    finally:
        dg.free_compressed(obj)


def test_grouped_splits_range_fallback_and_ld(mx):
    """a column beyond the denormal-operand range: the gated plain-operand pass redoes every group; padded Ldb / Ldc with poison"""
    o = Oracle()
    snps, indiv, n = 30_000, 2_500, 8
    prob = make_problem(snps, indiv, n, seed=77)
    dg = mx.dgemm_compressed
    L = mx.check_library_handle()
    dg.set_options(use_gpu=True, not_center=False, verbose=0)
    obj = dg.init_compressed(prob["plink"], prob["plink_t"], snps, indiv, prob["f"], n)
    try:
        for trans in (0, 1):
            k, m = (indiv, snps) if trans else (snps, indiv)
            ldb, ldc = k + 3, m + 5
            B8 = make_B(k, n, seed=6 + trans, ldb=ldb)
            B8[2, :k:3] *= 1e-300
            B8[2, 1:k:3] *= 1e+250
            outs = []
            for mb in (None, 1):
                C8 = np.full((n, ldc), -777.0)
                ctx = _env(MXA_P_BUDGET_MB=mb) if mb else _env()
                with ctx:
                    L.dgemm_compressed(b"T" if trans else b"N", obj, n, B8.ctypes.data_as(ctypes.c_void_p), ldb, C8.ctypes.data_as(ctypes.c_void_p), ldc)
                assert L.mxa_last_range_fallback(obj) == 1 and np.all(C8[:, m:] == 0.0)
                outs.append(C8)
            assert np.array_equal(outs[0], outs[1])
            ref8 = o.dgemm_dense(trans, prob, B8, 1, ldc=ldc)
            a8 = o.dgemm_dense(trans, prob, np.abs(B8), 0, ldc=ldc)[:, :m]
            assert np.all(np.abs(outs[1][:, :m] - ref8[:, :m]) <= k * 2.0 ** -52 * a8 + 8 * 2.0 ** -53 * np.abs(ref8[:, :m]) + 1e-300)
    finally:
        dg.free_compressed(obj)


def test_incremental_staging_equals_one_pointer_object(mx, tmp_path):
    import torch
    o = Oracle()
    snps, indiv, n = 5_003, 1_301, 12
    prob = make_problem(snps, indiv, n, seed=31, missing_frac=0.05)
    dg = mx.dgemm_compressed
    L = mx.check_library_handle()
    dg.set_options(use_gpu=True, not_center=False, verbose=0)
    with _env(MXA_SINGLE_ORIENTATION=1):
        ref_obj = dg.init_compressed(prob["plink"], None, snps, indiv, prob["f"], n)
    obj = dg.init_compressed_begin(snps, indiv, n)
    try:
        assert L.mxa_single_orientation(obj) == 1
        B = make_B(snps, n, seed=1)
        with pytest.raises(RuntimeError, match="still being staged"):
            dg.dgemm_compressed_main(False, obj, np.asfortranarray(B.T), snps, indiv)
        assert L.mxa_last_error() == 19
        # blocks out of order; host and device memory; frequencies given for one block, counted on the device for the others
        cuts = [0, 1000, 1004, 2500, 4096, snps]
        f_dev = mx.read_plink.calc_freq(prob["plink"], snps, indiv)
        for bi in (3, 0, 4, 1, 2):
            b0, b1 = cuts[bi], cuts[bi + 1]
            rows = np.ascontiguousarray(prob["plink"][b0:b1])
            if bi % 2:
                rows = torch.from_numpy(rows).to("cuda:0")
            dg.append_rows(obj, rows, b0, freq=f_dev[b0:b1] if bi == 4 else None)
        with pytest.raises(RuntimeError):                        # a row range outside the object
            dg.append_rows(obj, np.ascontiguousarray(prob["plink"][:8]), snps - 4)
        f = dg.init_compressed_end(obj, snps)
        assert np.array_equal(f, f_dev)
        with pytest.raises(RuntimeError):                        # sealed
            dg.append_rows(obj, np.ascontiguousarray(prob["plink"][:8]), 0)
        prob_dev_f = dict(prob, f=f_dev)                         # missing codes: the device count treats 01 as 0 alleles = the matrix the multiply uses
        dg.free_compressed(ref_obj)
        with _env(MXA_SINGLE_ORIENTATION=1):
            ref_obj = dg.init_compressed(prob["plink"], None, snps, indiv, f_dev, n)
        for trans in (0, 1):
            k, m = (indiv, snps) if trans else (snps, indiv)
            for nn in (12, 1, 5):
                Bt = make_B(k, nn, seed=2 + trans + nn)
                C = dg.dgemm_compressed_main(bool(trans), obj, np.asfortranarray(Bt.T), snps, indiv)
                Cr = dg.dgemm_compressed_main(bool(trans), ref_obj, np.asfortranarray(Bt.T), snps, indiv)
                assert np.array_equal(C, Cr)
                ref = o.dgemm_dense(trans, prob_dev_f, Bt, 1)[:, :m]
                assert np.abs(C.T - ref).max() <= RTOL * np.abs(ref).max()
        # an incomplete object cannot be sealed
        part = dg.init_compressed_begin(snps, indiv, n)
        dg.append_rows(part, np.ascontiguousarray(prob["plink"][:100]), 0)
        with pytest.raises(RuntimeError, match="100 SNP rows were appended"):
            dg.init_compressed_end(part, snps)
        dg.free_compressed(part)
        # coverage is tracked by interval, not by count: a block appended twice or overlapping another one is refused with error 1, so a
        # retried block cannot stand in for rows that were never written; and the frequencies of an unsealed object are not handed out
        part = dg.init_compressed_begin(snps, indiv, n)
        dg.append_rows(part, np.ascontiguousarray(prob["plink"][:100]), 0)
        for b0, nr in ((0, 100), (50, 100), (99, 1), (0, snps)):
            with pytest.raises(RuntimeError, match="overlap"):
                dg.append_rows(part, np.ascontiguousarray(prob["plink"][b0:b0 + nr]), b0)
        assert mx.lib.last_error()[0] == 1
        dg.append_rows(part, np.ascontiguousarray(prob["plink"][200:snps]), 200)          # rows [100, 200) are still missing: snps - 100 rows appended
        with pytest.raises(RuntimeError, match=r"first row not yet appended: 100"):
            dg.init_compressed_end(part, snps)
        fz = np.full(snps, -1.0)
        L.get_compressed_freq(part, mx.lib.ptr(fz))
        assert mx.lib.last_error()[0] == 19 and np.all(fz == -1.0)
        dg.append_rows(part, np.ascontiguousarray(prob["plink"][100:200]), 100)
        assert np.array_equal(dg.init_compressed_end(part, snps), f_dev)
        dg.free_compressed(part)
        # the .bed reader streams into a single-orientation object: same results, same frequencies
        bed = tmp_path / "x.bed"
        with open(bed, "wb") as fh:
            fh.write(bytes([0x6c, 0x1b, 0x01]))
            fh.write(prob["plink"].tobytes())
        with _env(MXA_SINGLE_ORIENTATION=1):
            bobj, bf, s_, i_ = dg.init_compressed_from_bed(str(bed), n, snps=snps, indiv=indiv)
        try:
            assert L.mxa_single_orientation(bobj) == 1 and np.array_equal(bf, f_dev)
            Bt = make_B(snps, n, seed=9)
            assert np.array_equal(dg.dgemm_compressed_main(False, bobj, np.asfortranarray(Bt.T), snps, indiv),
                                  dg.dgemm_compressed_main(False, ref_obj, np.asfortranarray(Bt.T), snps, indiv))
            with open(tmp_path / "short.bed", "wb") as fh:      # a truncated file is an error, no object
                fh.write(bytes([0x6c, 0x1b, 0x01]))
                fh.write(prob["plink"].tobytes()[:-5])
            with _env(MXA_SINGLE_ORIENTATION=1), pytest.raises(RuntimeError):
                dg.init_compressed_from_bed(str(tmp_path / "short.bed"), n, snps=snps, indiv=indiv)
        finally:
            dg.free_compressed(bobj)
    finally:
        dg.free_compressed(obj)
        dg.free_compressed(ref_obj)
