"""Several SNP shards BEHIND the reference C ABI (mxa_multi.cpp): MIRACULIX_NUM_GPUS > visible devices puts the shards onto the one
GPU of the test box as "virtual shards" -- the same host code, worker threads, per-shard streams, partial buffers and fixed-order
reduction as on a multi-GPU node, only the peer copies are local.  Checked against the oracle, against the single-device object
(exactly, for integer-valued operands) and for run-to-run bitwise reproducibility.  Also: shard staging from a .bed file by SNP
range (mxa_bed2compressed_range), and the RCCL reduction with a one-rank communicator."""
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
        self.kv = kv

    def __enter__(self):
        self.old = {k: os.environ.get(k) for k in self.kv}
        for k, v in self.kv.items():
            os.environ[k] = str(v)

    def __exit__(self, *a):
        for k, v in self.old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def _make(mx, prob, n, shards, **extra):
    dg = mx.dgemm_compressed
    with _env(MIRACULIX_NUM_GPUS=shards, **extra):
        obj = dg.init_compressed(prob["plink"], prob["plink_t"], prob["snps"], prob["indiv"], prob["f"], n)
    return obj


@pytest.mark.parametrize("shards", [2, 3, 8])
@pytest.mark.parametrize("centered", [0, 1])
def test_multi_object_matches_oracle_host_operands(mx, shards, centered):
    o = Oracle()
    snps, indiv, n = 2051, 777, 6
    prob = make_problem(snps, indiv, n, seed=8)
    dg = mx.dgemm_compressed
    dg.set_options(use_gpu=True, not_center=not centered, verbose=0)
    obj = _make(mx, prob, n, shards)
    try:
        assert dg.num_shards(obj) == shards
        BN, BT = make_B(snps, n, seed=1), make_B(indiv, n, seed=2)
        refN = o.dgemm_dense(0, prob, BN, centered)
        refT = o.dgemm_dense(1, prob, BT, centered)
        CN = dg.dgemm_compressed_main(False, obj, np.asfortranarray(BN.T), snps, indiv)
        CT = dg.dgemm_compressed_main(True, obj, np.asfortranarray(BT.T), snps, indiv)
        assert np.abs(CN.T - refN).max() <= RTOL * np.abs(refN).max()
        assert np.abs(CT.T - refT).max() <= RTOL * np.abs(refT).max()
        # bitwise reproducible: fixed split-K order inside a shard, ascending shard order in the reduction
        assert np.array_equal(CN, dg.dgemm_compressed_main(False, obj, np.asfortranarray(BN.T), snps, indiv))
        # frequencies come back in SNP order
        L = mx.check_library_handle()
        fq = np.zeros(snps)
        L.get_compressed_freq(obj, fq.ctypes.data_as(ctypes.c_void_p))
        assert np.array_equal(fq, prob["f"])
    finally:
        dg.free_compressed(obj)
    assert obj.value is None


def test_multi_object_raw_abi_ld_padding_and_device_operands(mx):
    """padded Ldb (poisoned) and Ldc (zero-filled) through the raw symbols; then the same with device-resident B / C; gram step"""
    import torch
    o = Oracle()
    L = mx.check_library_handle()
    snps, indiv, n = 3001, 517, 5
    prob = make_problem(snps, indiv, n, seed=21)
    dg = mx.dgemm_compressed
    dg.set_options(use_gpu=True, not_center=False, verbose=0)
    obj = _make(mx, prob, n, 4)
    single = dg.init_compressed(prob["plink"], prob["plink_t"], snps, indiv, prob["f"], n)
    try:
        for trans in (0, 1):
            k = indiv if trans else snps
            m = snps if trans else indiv
            ldb, ldc = k + 3, m + 5
            B = make_B(k, n, seed=4 + trans, ldb=ldb)                       # n x ldb, padding poisoned with 1e300
            ref = o.dgemm_dense(trans, prob, B, 1)[:, :m]
            C = np.full((n, ldc), -777.0)
            L.dgemm_compressed(b"T" if trans else b"N", obj, n, B.ctypes.data_as(ctypes.c_void_p), ldb, C.ctypes.data_as(ctypes.c_void_p), ldc)
            assert L.mxa_last_error() == 0
            assert np.abs(C[:, :m] - ref).max() <= RTOL * np.abs(ref).max()
            assert np.all(C[:, m:] == 0.0)
            # device operands (on the device that also holds the shards)
            Bd = torch.from_numpy(B).cuda()
            Cd = torch.full((n, ldc), -777.0, dtype=torch.float64, device="cuda")
            L.dgemm_compressed(b"T" if trans else b"N", obj, n, ctypes.c_void_p(Bd.data_ptr()), ldb, ctypes.c_void_p(Cd.data_ptr()), ldc)
            assert L.mxa_last_error() == 0
            assert np.array_equal(Cd.cpu().numpy(), C)
        V = make_B(indiv, n, seed=9)
        G = dg.gram_matvec(obj, np.asfortranarray(V.T), snps, indiv)
        t_ref = o.dgemm_dense(1, prob, V, 1)[:, :snps]
        ref = o.dgemm_dense(0, prob, np.ascontiguousarray(t_ref), 1)[:, :indiv].T
        assert np.abs(G - ref).max() <= RTOL * np.abs(ref).max()
        # integer-valued operands: every partial sum is exact, so the sharded object equals the single-device object bit for bit
        dg.set_options(use_gpu=True, not_center=True, verbose=0)
        rng = np.random.default_rng(3)
        Bi = np.asfortranarray(rng.integers(-50, 50, size=(snps, n)).astype(np.float64))
        assert np.array_equal(dg.dgemm_compressed_main(False, obj, Bi, snps, indiv), dg.dgemm_compressed_main(False, single, Bi, snps, indiv))
        Bi = np.asfortranarray(rng.integers(-50, 50, size=(indiv, n)).astype(np.float64))
        assert np.array_equal(dg.dgemm_compressed_main(True, obj, Bi, snps, indiv), dg.dgemm_compressed_main(True, single, Bi, snps, indiv))
        # the asynchronous single-device entry is refused on a multi-device handle
        assert L.mxa_dgemm_compressed_device(b"N", obj, 1, None, snps, None, indiv, None, 1) == 1
    finally:
        dg.free_compressed(obj)
        dg.free_compressed(single)


def test_more_shards_than_snp_quadruples(mx):
    """4 * shards > snps: empty blocks are dropped, the rest still covers every SNP"""
    o = Oracle()
    snps, indiv, n = 10, 37, 2
    prob = make_problem(snps, indiv, n, seed=5)
    dg = mx.dgemm_compressed
    dg.set_options(use_gpu=True, not_center=False, verbose=0)
    obj = _make(mx, prob, n, 8)
    try:
        assert dg.num_shards(obj) == 3
        for trans in (0, 1):
            B = make_B(indiv if trans else snps, n, seed=1)
            ref = o.dgemm_dense(trans, prob, B, 1)[:, : (snps if trans else indiv)]
            C = dg.dgemm_compressed_main(bool(trans), obj, np.asfortranarray(B.T), snps, indiv)
            assert np.abs(C.T - ref).max() <= RTOL * np.abs(ref).max()
    finally:
        dg.free_compressed(obj)


def test_rccl_reduction_one_rank(mx):
    """MXA_REDUCE=rccl with a one-shard multi object: ncclCommInitAll / ncclReduce are bound from librccl.so at run time and run
    (one rank is all a one-GPU box allows; more shards than devices fall back to the peer-to-peer reduction, also checked)"""
    o = Oracle()
    snps, indiv, n = 1200, 333, 3
    prob = make_problem(snps, indiv, n, seed=6)
    dg = mx.dgemm_compressed
    dg.set_options(use_gpu=True, not_center=False, verbose=0)
    B = make_B(snps, n, seed=1)
    ref = o.dgemm_dense(0, prob, B, 1)[:, :indiv]
    for shards, extra in ((1, dict(MXA_FORCE_MULTI=1, MXA_REDUCE="rccl")), (2, dict(MXA_REDUCE="rccl"))):
        obj = _make(mx, prob, n, shards, **extra)
        try:
            assert dg.num_shards(obj) == shards
            C = dg.dgemm_compressed_main(False, obj, np.asfortranarray(B.T), snps, indiv)
            assert np.abs(C.T - ref).max() <= RTOL * np.abs(ref).max()
        finally:
            dg.free_compressed(obj)


def test_bed_ranges_reproduce_the_unsharded_object(mx, tmp_path):
    """three SNP ranges staged straight from the .bed file (only their rows are read) against the object staged from the whole
    file: frequencies bit for bit, products bit for bit for integer-valued operands (all sums exact), within the stated tolerance
    for random ones; the in-process sharder over the same file (MIRACULIX_NUM_GPUS=3) likewise"""
    o = Oracle()
    snps, indiv, n = 2050, 613, 4
    prob = make_problem(snps, indiv, n, seed=31)
    bed = str(tmp_path / "x.bed")
    mx.read_plink.write_bed(bed, prob["plink"])
    dg = mx.dgemm_compressed
    dg.set_options(use_gpu=True, not_center=False, verbose=0)
    full, f_full, s_out, i_out = dg.init_compressed_from_bed(bed, n, snps=snps, indiv=indiv)
    assert (s_out, i_out) == (snps, indiv) and np.array_equal(f_full, prob["f"])
    cuts = [0, 684, 1370, snps]          # deliberately not all multiples of 4: a range object is self-contained
    parts = [dg.init_compressed_from_bed_range(bed, cuts[i], cuts[i + 1], n, snps=snps, indiv=indiv) for i in range(3)]
    with _env(MIRACULIX_NUM_GPUS=3):
        multi, f_multi, _, _ = dg.init_compressed_from_bed(bed, n, snps=snps, indiv=indiv)
    try:
        assert np.array_equal(np.concatenate([f for _, f in parts]), f_full)
        assert np.array_equal(f_multi, f_full) and dg.num_shards(multi) == 3
        rng = np.random.default_rng(1)
        for centered in (0, 1):
            dg.set_options(use_gpu=True, not_center=not centered, verbose=0)
            for integer in (True, False):
                BN = rng.integers(-9, 9, size=(snps, n)).astype(np.float64) if integer else rng.standard_normal((snps, n))
                BT = rng.integers(-9, 9, size=(indiv, n)).astype(np.float64) if integer else rng.standard_normal((indiv, n))
                BN, BT = np.asfortranarray(BN), np.asfortranarray(BT)
                CN = dg.dgemm_compressed_main(False, full, BN, snps, indiv)
                CT = dg.dgemm_compressed_main(True, full, BT, snps, indiv)
                PN = sum(dg.dgemm_compressed_main(False, obj, np.asfortranarray(BN[cuts[i]:cuts[i + 1]]), cuts[i + 1] - cuts[i], indiv) for i, (obj, _) in enumerate(parts))
                PT = np.concatenate([dg.dgemm_compressed_main(True, obj, BT, cuts[i + 1] - cuts[i], indiv) for i, (obj, _) in enumerate(parts)])
                MN = dg.dgemm_compressed_main(False, multi, BN, snps, indiv)
                MT = dg.dgemm_compressed_main(True, multi, BT, snps, indiv)
                if integer and not centered:
                    assert np.array_equal(PN, CN) and np.array_equal(PT, CT) and np.array_equal(MN, CN) and np.array_equal(MT, CT)
                else:
                    for X, Y in ((PN, CN), (PT, CT), (MN, CN), (MT, CT)):
                        assert np.abs(X - Y).max() <= RTOL * np.abs(Y).max()
            ref = o.dgemm_dense(0, prob, np.ascontiguousarray(BN.T), centered)[:, :indiv]
            assert np.abs(MN.T - ref).max() <= RTOL * np.abs(ref).max()
    finally:
        dg.free_compressed(full)
        dg.free_compressed(multi)
        for obj, _ in parts:
            dg.free_compressed(obj)


def test_multi_object_with_pipelined_host_transfers(mx):
    """two shards whose 'T' result blocks (150k x 32 doubles = 38 MB each) are large enough for the row-range pipeline of
    gemm_host_pipelined: it then runs inside both worker threads at the same time; integer-valued operands make every sum exact, so the
    sharded object must equal the single-device object bit for bit"""
    import torch
    from bench import synth_genotypes_device
    dev = torch.device("cuda", 0)
    snps, indiv, n = 300_000, 2_001, 32
    plink = synth_genotypes_device(torch, snps, indiv, 9, dev)
    plink_t = mx.compressed_operations.transpose_genotype_matrix(plink, snps, indiv)
    f = mx.read_plink.calc_freq(plink, snps, indiv)
    dg = mx.dgemm_compressed
    dg.set_options(use_gpu=True, not_center=True, verbose=0)
    single = dg.init_compressed(plink, plink_t, snps, indiv, f, n)
    with _env(MIRACULIX_NUM_GPUS=2):
        multi = dg.init_compressed(plink, plink_t, snps, indiv, f, n)
    try:
        assert dg.num_shards(multi) == 2
        rng = np.random.default_rng(4)
        BT = np.asfortranarray(rng.integers(-20, 20, size=(indiv, n)).astype(np.float64))
        BN = np.asfortranarray(rng.integers(-20, 20, size=(snps, n)).astype(np.float64))
        assert np.array_equal(dg.dgemm_compressed_main(True, multi, BT, snps, indiv), dg.dgemm_compressed_main(True, single, BT, snps, indiv))
        assert np.array_equal(dg.dgemm_compressed_main(False, multi, BN, snps, indiv), dg.dgemm_compressed_main(False, single, BN, snps, indiv))
    finally:
        dg.free_compressed(multi)
        dg.free_compressed(single)


def test_multi_object_life_cycle_does_not_leak_device_memory(mx):
    """create / multiply / free 40 three-shard objects with host operands (strided downloads from three worker threads at once): device memory
    must return to where it was -- concurrent hipMemcpy2DAsync downloads left 0.2-0.6 MiB per shard behind before they went per column"""
    import torch
    snps, indiv, n = 6001, 1201, 10
    prob = make_problem(snps, indiv, n, seed=3)
    dg = mx.dgemm_compressed
    dg.set_options(use_gpu=True, not_center=True, verbose=0)
    rng = np.random.default_rng(0)
    BT = np.asfortranarray(rng.standard_normal((indiv, n)))
    BN = np.asfortranarray(rng.standard_normal((snps, n)))

    def free_bytes():
        torch.cuda.synchronize()
        return torch.cuda.mem_get_info()[0]

    base = None
    for it in range(48):
        obj = _make(mx, prob, n, 3)
        dg.dgemm_compressed_main(True, obj, BT, snps, indiv)
        dg.dgemm_compressed_main(False, obj, BN, snps, indiv)
        dg.dgemm_compressed_main(True, obj, np.asfortranarray(BT[:, :1]), snps, indiv)
        dg.free_compressed(obj)
        if it == 7:
            base = free_bytes()
    assert base - free_bytes() <= 8 << 20


@pytest.mark.parametrize("rows,parts", [(1000, 3), (300, 4), (2100, 2)])
def test_crossproduct_column_panels_inside_one_process(mx, rows, parts):
    """snp_multiply_gpu with MIRACULIX_NUM_GPUS > 1 and host operands: every 'device' computes a column panel of the symmetric result and
    downloads it into its slab of the host matrix (virtual devices on a one-GPU box); bit-identical to the single-device call and to the oracle"""
    o = Oracle()
    k = 4099
    rng = np.random.default_rng(rows)
    X = rng.integers(0, 256, size=(rows, (k + 3) // 4), dtype=np.uint8)
    X[:, -1] &= (1 << (2 * (k % 4))) - 1
    single = mx.crossproduct.snp_crossprod(X, k, rows, is_snpmajor=False, is_plink_format=True)
    with _env(MIRACULIX_NUM_GPUS=parts):
        multi = mx.crossproduct.snp_crossprod(X, k, rows, is_snpmajor=False, is_plink_format=True)
    assert np.array_equal(multi, single)
    assert np.array_equal(multi, o.crossprod_i32(X, k, True).astype(np.float64))
