"""GPU parity for the on-device staging helpers (2-bit transpose, popcount frequencies) and for operands that already
live in HBM (torch device tensors through the same C ABI).  Mirrors tests/dgemm_compressed/test.jl:60-83."""
import numpy as np
import pytest

from _util import Oracle, make_B, make_problem

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mx():
    import miraculix_amd as m
    m.load_shared_library()
    return m


# the last six shapes have both row pitches multiples of 4 bytes: the tiled butterfly kernel (256 x 512 tiles, ragged edges, partial
# last dwords); the others take the generic kernel
@pytest.mark.parametrize("snps,indiv", [(1000, 500), (1003, 501), (64, 64), (130, 1027), (4097, 255),
                                        (16, 16), (1024, 512), (1022, 509), (2064, 784), (272, 4112), (6000, 4000)])
def test_transpose_and_freq(mx, snps, indiv):
    o = Oracle()
    prob = make_problem(snps, indiv, 1, seed=snps)
    T = mx.compressed_operations.transpose_genotype_matrix(prob["plink"], snps, indiv)
    assert np.array_equal(T, prob["plink_t"])
    assert np.array_equal(T, o.transpose_2bit(prob["plink"], snps, indiv))
    # round trip through decompression like test.jl:79-83
    Zt = mx.compressed_operations.decompress_genotype_matrix(T, indiv, snps)
    assert np.array_equal(Zt, prob["Z"])
    f = mx.read_plink.calc_freq(prob["plink"], snps, indiv)
    assert np.array_equal(f, o.allele_freq(prob["plink"], snps, indiv))
    assert np.abs(f - prob["f"]).max() < 1e-15


def test_device_resident_operands(mx):
    import torch
    o = Oracle()
    dev = torch.device("cuda", 0)
    prob = make_problem(2051, 1030, 12, seed=5)
    dg = mx.dgemm_compressed
    dg.set_options(use_gpu=True, not_center=False, verbose=0)
    plink_d = torch.from_numpy(prob["plink"]).to(dev)
    plink_t_d = mx.compressed_operations.transpose_genotype_matrix(plink_d, prob["snps"], prob["indiv"])
    assert np.array_equal(plink_t_d.cpu().numpy(), prob["plink_t"])
    f_d = mx.read_plink.calc_freq(plink_d, prob["snps"], prob["indiv"])
    obj = dg.init_compressed(plink_d, plink_t_d, prob["snps"], prob["indiv"], f_d, 12)
    for trans in (0, 1):
        k = prob["indiv"] if trans else prob["snps"]
        m = prob["snps"] if trans else prob["indiv"]
        ldb = k + 3
        B = make_B(k, 12, seed=9, ldb=ldb)              # (n x ldb) rows = columns, poisoned padding
        B_d = torch.from_numpy(B).to(dev).t()[:k, :]     # k x n view, column stride ldb
        C_d = dg.dgemm_compressed_main(bool(trans), obj, B_d, prob["snps"], prob["indiv"])
        ref = o.dgemm_dense(trans, prob, B, 1)[:, :m]
        err = np.abs(C_d.t().cpu().numpy() - ref).max() / np.abs(ref).max()
        assert err <= 1e-11, err
    dg.free_compressed(obj)


def test_ldc_padding_zero_filled_and_ldb_honoured(mx):
    """the reference CPU path honours Ldb/Ldc and zero-fills the padding rows of C (5codesIntern.h:67); so does this"""
    import ctypes
    o = Oracle()
    prob = make_problem(1200, 333, 3, seed=8)
    dg = mx.dgemm_compressed
    L = mx.check_library_handle()
    dg.set_options(use_gpu=True, not_center=False, verbose=0)
    obj = dg.init_compressed(prob["plink"], prob["plink_t"], 1200, 333, prob["f"], 3)
    for trans in (0, 1):
        k = 333 if trans else 1200
        m = 1200 if trans else 333
        ldb, ldc = k + 5, m + 7
        B = make_B(k, 3, seed=4, ldb=ldb)
        C = np.full((3, ldc), -777.0)
        L.dgemm_compressed(b"T" if trans else b"N", obj, 3, B.ctypes.data_as(ctypes.c_void_p), ldb, C.ctypes.data_as(ctypes.c_void_p), ldc)
        ref = o.dgemm_dense(trans, prob, B, 1, ldc=ldc)
        assert np.all(C[:, m:] == 0.0)
        assert np.abs(C - ref).max() / np.abs(ref).max() <= 1e-11
    dg.free_compressed(obj)


def test_bed_file_staging(mx, tmp_path):
    """mxa_bed2compressed: host C++ reads the .bed (magic 6c 1b 01), the device builds the transposed copy and the frequencies;
    dimensions come from the .bim/.fam line counts (read_plink.jl:161-222)"""
    o = Oracle()
    prob = make_problem(1237, 415, 4, seed=17)
    base = str(tmp_path / "toy")
    mx.read_plink.write_bed(base + ".bed", prob["plink"])
    with open(base + ".bim", "w") as fh:
        fh.write("".join(f"1 snp{i} 0 {i} A B\n" for i in range(1237)))
    with open(base + ".fam", "w") as fh:
        fh.write("".join(f"f{i} i{i} 0 0 0 -9\n" for i in range(415)))
    dg = mx.dgemm_compressed
    dg.set_options(use_gpu=True, not_center=False, verbose=0)
    obj, f, snps, indiv = dg.init_compressed_from_bed(base + ".bed", 4)
    assert (snps, indiv) == (1237, 415)
    assert np.array_equal(f, o.allele_freq(prob["plink"], 1237, 415))
    for trans in (0, 1):
        k = indiv if trans else snps
        m = snps if trans else indiv
        B = make_B(k, 4, seed=2)
        C = dg.dgemm_compressed_main(bool(trans), obj, np.asfortranarray(B.T), snps, indiv)
        ref = o.dgemm_dense(trans, prob, B, 1)[:, :m]
        assert np.abs(C.T - ref).max() <= 1e-11 * np.abs(ref).max()
    dg.free_compressed(obj)
    # python-side reader agrees
    p2, s2, i2 = mx.read_plink.read_bed(base + ".bed")
    assert (s2, i2) == (1237, 415) and np.array_equal(p2, prob["plink"])


@pytest.mark.parametrize("snps,indiv,missing", [(1003, 501, 0.0), (2064, 784, 0.1), (130, 1027, 0.05), (4097, 255, 0.0)])
@pytest.mark.parametrize("how", ["null", "same_pointer", "device_source"])
def test_one_pointer_staging_is_bit_identical(mx, snps, indiv, missing, how):
    """plink2compressed with plink_transposed == NULL or == plink (the reference's CPU call shape: 5codesChar.cc:368-393 never reads the
    transposed matrix, utils/benchmark/benchmark.f90:185 passes plink twice): the individual-major copy is built on the device from the
    raw PLINK codes, so both products are bit-identical to those of the object staged from two pointers -- missing codes (01) included --
    and agree with the oracle."""
    import torch
    o = Oracle()
    n = 9
    prob = make_problem(snps, indiv, n, seed=snps + 1, missing_frac=missing)
    dg = mx.dgemm_compressed
    dg.set_options(use_gpu=True, not_center=False, verbose=0)
    ref_obj = dg.init_compressed(prob["plink"], prob["plink_t"], snps, indiv, prob["f"], n)
    if how == "null":
        obj = dg.init_compressed(prob["plink"], None, snps, indiv, prob["f"], n)
    elif how == "same_pointer":
        obj = dg.init_compressed(prob["plink"], prob["plink"], snps, indiv, prob["f"], n)
    else:
        obj = dg.init_compressed(torch.from_numpy(prob["plink"]).cuda(), None, snps, indiv, prob["f"], n)
    try:
        for trans in (0, 1):
            k, m = (indiv, snps) if trans else (snps, indiv)
            B = make_B(k, n, seed=3 + trans)
            C2 = dg.dgemm_compressed_main(bool(trans), ref_obj, np.asfortranarray(B.T), snps, indiv)
            C1 = dg.dgemm_compressed_main(bool(trans), obj, np.asfortranarray(B.T), snps, indiv)
            assert np.array_equal(C1, C2)
            ref = o.dgemm_dense(trans, prob, B, 1)[:, :m]
            assert np.abs(C1.T - ref).max() <= 1e-11 * np.abs(ref).max()
    finally:
        dg.free_compressed(obj)
        dg.free_compressed(ref_obj)


def test_one_pointer_staging_under_num_gpus(mx, monkeypatch):
    """the same call shape with MIRACULIX_NUM_GPUS = 3 (virtual shards on one device): every shard transposes its own SNP block"""
    o = Oracle()
    snps, indiv, n = 3001, 777, 5
    prob = make_problem(snps, indiv, n, seed=21)
    dg = mx.dgemm_compressed
    dg.set_options(use_gpu=True, not_center=False, verbose=0)
    monkeypatch.setenv("MIRACULIX_NUM_GPUS", "3")
    obj = dg.init_compressed(prob["plink"], None, snps, indiv, prob["f"], n)
    monkeypatch.delenv("MIRACULIX_NUM_GPUS")
    try:
        assert dg.num_shards(obj) == 3
        for trans in (0, 1):
            k, m = (indiv, snps) if trans else (snps, indiv)
            B = make_B(k, n, seed=8 + trans)
            C = dg.dgemm_compressed_main(bool(trans), obj, np.asfortranarray(B.T), snps, indiv)
            ref = o.dgemm_dense(trans, prob, B, 1)[:, :m]
            assert np.abs(C.T - ref).max() <= 1e-11 * np.abs(ref).max()
    finally:
        dg.free_compressed(obj)
