"""Parity at BASELINE.json's full single-GPU size (1M SNPs x 50k individuals) through size-independent properties, because the
oracle cannot run there: exact checksums (B = ones reproduces integer row/column sums that the popcount kernel gives
independently), the adjoint identity <x, Z y> = <Z^T x, y> tying the two stored orientations together, linearity, and
agreement of a random row/column sample with the dense oracle.  Exercises 64-bit addressing (12.5 GB per orientation)."""
import ctypes

import numpy as np
import pytest

from _util import Oracle, elementwise_bound

pytestmark = pytest.mark.gpu

SNPS, INDIV, N = 1_000_000, 50_000, 32


@pytest.fixture(scope="module")
def big():
    import torch
    import miraculix_amd as mx
    from bench import synth_plink_device
    mx.load_shared_library()
    dev = torch.device("cuda", 0)
    plink = synth_plink_device(torch, SNPS, (INDIV + 3) // 4, 42, dev)
    plink_t = mx.compressed_operations.transpose_genotype_matrix(plink, SNPS, INDIV)
    f = mx.read_plink.calc_freq(plink, SNPS, INDIV)
    dg = mx.dgemm_compressed
    dg.set_options(use_gpu=True, not_center=True, verbose=0)
    obj = dg.init_compressed(plink, plink_t, SNPS, INDIV, f, N)
    yield dict(mx=mx, torch=torch, dev=dev, plink=plink, plink_t=plink_t, f=f, obj=obj, dg=dg)
    dg.free_compressed(obj)


def test_checksums_exact(big):
    """B = 1: 'T' gives per-SNP allele counts = 2*indiv*f_s, 'N' gives per-individual allele counts; both are integers and
    must match the independent popcount kernel exactly"""
    torch, mx, dg = big["torch"], big["mx"], big["dg"]
    ones_i = torch.ones((INDIV, 1), dtype=torch.float64, device=big["dev"])
    ct = dg.dgemm_compressed_main(True, big["obj"], ones_i, SNPS, INDIV)          # snps x 1
    counts_s = torch.round(big["f"] * (2.0 * INDIV))
    assert torch.equal(ct[:, 0], counts_s)
    ones_s = torch.ones((SNPS, 1), dtype=torch.float64, device=big["dev"])
    cn = dg.dgemm_compressed_main(False, big["obj"], ones_s, SNPS, INDIV)         # indiv x 1
    fi = mx.read_plink.calc_freq(big["plink_t"], INDIV, SNPS)                      # the transposed matrix read as "INDIV SNPs"
    assert torch.equal(cn[:, 0], torch.round(fi * (2.0 * SNPS)))
    assert float(cn.sum()) == float(ct.sum())


def test_adjoint_identity_and_linearity(big):
    torch, dg = big["torch"], big["dg"]
    g = torch.Generator(device=big["dev"]); g.manual_seed(3)
    Y = torch.randn((N, SNPS), dtype=torch.float64, device=big["dev"], generator=g).t()      # snps x n
    X = torch.randn((N, INDIV), dtype=torch.float64, device=big["dev"], generator=g).t()     # indiv x n
    ZY = dg.dgemm_compressed_main(False, big["obj"], Y, SNPS, INDIV)                          # indiv x n
    ZtX = dg.dgemm_compressed_main(True, big["obj"], X, SNPS, INDIV)                          # snps x n
    lhs = (X * ZY).sum(dim=0)
    rhs = (ZtX * Y).sum(dim=0)
    assert float(((lhs - rhs).abs() / lhs.abs().clamp_min(1.0)).max()) <= 1e-10
    # linearity: Z (Y + 3 Y2) = Z Y + 3 Z Y2
    Y2 = torch.randn((N, SNPS), dtype=torch.float64, device=big["dev"], generator=g).t()
    ZY2 = dg.dgemm_compressed_main(False, big["obj"], Y2, SNPS, INDIV)
    ZS = dg.dgemm_compressed_main(False, big["obj"], (Y + 3.0 * Y2).t().contiguous().t(), SNPS, INDIV)
    assert float((ZS - (ZY + 3.0 * ZY2)).abs().max()) <= 1e-11 * float(ZS.abs().max())
    # bitwise reproducible (fixed split-K order, no atomics)
    ZY_again = dg.dgemm_compressed_main(False, big["obj"], Y, SNPS, INDIV)
    assert torch.equal(ZY, ZY_again)


def test_sampled_rows_vs_dense_oracle(big):
    """64 individuals and 64 SNPs of the full-size results against the long-double dense oracle on the extracted sub-matrices"""
    torch, dg = big["torch"], big["dg"]
    o = Oracle()
    g = torch.Generator(device=big["dev"]); g.manual_seed(5)
    rng = np.random.default_rng(1)
    # 'N': rows = individuals; oracle needs those individuals' full SNP rows -> build a 64-individual problem
    ii = np.sort(rng.choice(INDIV, 64, replace=False))
    Y = torch.randn((4, SNPS), dtype=torch.float64, device=big["dev"], generator=g).t()
    ZY = dg.dgemm_compressed_main(False, big["obj"], Y, SNPS, INDIV)
    rows = big["plink_t"][torch.from_numpy(ii).to(big["dev"])].cpu().numpy()                 # 64 x ceil(snps/4)
    sub_plink = o.transpose_2bit(np.ascontiguousarray(rows), 64, SNPS)                        # snps x 16 bytes
    prob = dict(snps=SNPS, indiv=64, plink=sub_plink, plink_t=rows, f=np.zeros(SNPS))
    Yh = np.ascontiguousarray(Y.t().cpu().numpy())
    ref = o.dgemm_dense(0, prob, Yh, 0)                                                       # 4 x 64
    got = ZY[torch.from_numpy(ii).to(big["dev"])].t().cpu().numpy()
    assert np.abs(got - ref).max() <= 1e-11 * np.abs(ref).max()
    assert np.all(np.abs(got - ref) <= elementwise_bound(o, 0, prob, Yh, 0))                  # per element: 4 K 2^-53 sum |z||b|
    # 'T': rows = SNPs
    ss = np.sort(rng.choice(SNPS, 64, replace=False))
    X = torch.randn((4, INDIV), dtype=torch.float64, device=big["dev"], generator=g).t()
    ZtX = dg.dgemm_compressed_main(True, big["obj"], X, SNPS, INDIV)
    srows = big["plink"][torch.from_numpy(ss).to(big["dev"])].cpu().numpy()                   # 64 x ceil(indiv/4)
    prob = dict(snps=64, indiv=INDIV, plink=np.ascontiguousarray(srows), plink_t=None, f=np.zeros(64))
    Xh = np.ascontiguousarray(X.t().cpu().numpy())
    ref = o.dgemm_dense(1, prob, Xh, 0)
    got = ZtX[torch.from_numpy(ss).to(big["dev"])].t().cpu().numpy()
    assert np.abs(got - ref).max() <= 1e-11 * np.abs(ref).max()
    assert np.all(np.abs(got - ref) <= elementwise_bound(o, 1, prob, Xh, 0))


def test_sampled_rows_fp64_mfma_path_elementwise(big):
    """the headline products themselves (n = 32: k_gemm on the fp64 MFMA, both forms) -- 48 sampled rows, four of the 32 columns, against the long-double oracle
    under the hard ELEMENT-WISE bound |C_ij - ref_ij| <= 4 K 2^-53 sum_k |z_ik| |b_kj|"""
    torch, dg = big["torch"], big["dg"]
    o = Oracle()
    g = torch.Generator(device=big["dev"]); g.manual_seed(21)
    rng = np.random.default_rng(4)
    cols = [0, 13, 30, 31]
    ii = np.sort(rng.choice(INDIV, 48, replace=False))
    Y = torch.randn((N, SNPS), dtype=torch.float64, device=big["dev"], generator=g).t()
    ZY = dg.dgemm_compressed_main(False, big["obj"], Y, SNPS, INDIV)
    assert dg.last_path() == "k_gemm"
    rows = big["plink_t"][torch.from_numpy(ii).to(big["dev"])].cpu().numpy()
    sub_plink = o.transpose_2bit(np.ascontiguousarray(rows), 48, SNPS)
    prob = dict(snps=SNPS, indiv=48, plink=sub_plink, plink_t=rows, f=np.zeros(SNPS))
    Yh = np.ascontiguousarray(Y[:, cols].t().cpu().numpy())
    ref = o.dgemm_dense(0, prob, Yh, 0)
    got = ZY[torch.from_numpy(ii).to(big["dev"])][:, cols].t().cpu().numpy()
    assert np.all(np.abs(got - ref) <= elementwise_bound(o, 0, prob, Yh, 0))
    del Y, ZY
    ss = np.sort(rng.choice(SNPS, 48, replace=False))
    X = torch.randn((N, INDIV), dtype=torch.float64, device=big["dev"], generator=g).t()
    ZtX = dg.dgemm_compressed_main(True, big["obj"], X, SNPS, INDIV)
    assert dg.last_path() == "k_gemm"
    srows = big["plink"][torch.from_numpy(ss).to(big["dev"])].cpu().numpy()
    prob = dict(snps=48, indiv=INDIV, plink=np.ascontiguousarray(srows), plink_t=None, f=np.zeros(48))
    Xh = np.ascontiguousarray(X[:, cols].t().cpu().numpy())
    ref = o.dgemm_dense(1, prob, Xh, 0)
    got = ZtX[torch.from_numpy(ss).to(big["dev"])][:, cols].t().cpu().numpy()
    assert np.all(np.abs(got - ref) <= elementwise_bound(o, 1, prob, Xh, 0))


def test_exact_int8_engine_at_full_size(big):
    """engine i8-exact at 1M x 50k x 32: taken (standard-normal B needs 10-11 digits), adjoint identity, agreement with the fp64 engine, and the stated
    element-wise bound |error| <= 3.02 (S - 1) 2^-53 sum_k |z_k b_k| on 32 sampled individuals against the long-double dense oracle"""
    import os
    torch, dg, mx = big["torch"], big["dg"], big["mx"]
    o = Oracle()
    # the opt-in engines at wide n multiply 'N' on the plain int8 kernel, which reads the individual-major copy: an object with BOTH copies
    # (MXA_SINGLE_ORIENTATION=0; the default object keeps the SNP-major copy alone and sends such an 'N' to the fp64 engine)
    os.environ["MXA_SINGLE_ORIENTATION"] = "0"
    try:
        obj2 = dg.init_compressed(big["plink"], big["plink_t"], SNPS, INDIV, big["f"], N)
    finally:
        os.environ.pop("MXA_SINGLE_ORIENTATION", None)
    big = dict(big, obj=obj2)
    try:
        _exact_int8_engine_body(big, torch, dg, mx, o)
    finally:
        dg.free_compressed(obj2)


def _exact_int8_engine_body(big, torch, dg, mx, o):
    g = torch.Generator(device=big["dev"]); g.manual_seed(11)
    Y = torch.randn((N, SNPS), dtype=torch.float64, device=big["dev"], generator=g).t()
    X = torch.randn((N, INDIV), dtype=torch.float64, device=big["dev"], generator=g).t()
    ZY64 = dg.dgemm_compressed_main(False, big["obj"], Y, SNPS, INDIV)
    assert dg.last_path() == "k_gemm"
    prev = dg.set_engine("i8-exact")
    try:
        ZY = dg.dgemm_compressed_main(False, big["obj"], Y, SNPS, INDIV)
        assert dg.last_path() == "k_gemm_i8"
        L = mx.check_library_handle()
        gm, gk, gn, gs, ga, gc = ctypes.c_long(), ctypes.c_long(), ctypes.c_int(), ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        L.mxa_last_geometry(ctypes.byref(gm), ctypes.byref(gk), ctypes.byref(gn), ctypes.byref(gs), ctypes.byref(ga), ctypes.byref(gc))
        S = ga.value
        assert 9 <= S <= 13
        ZtX = dg.dgemm_compressed_main(True, big["obj"], X, SNPS, INDIV)
        assert dg.last_path() == "k_gemm_i8"
        assert torch.equal(ZY, dg.dgemm_compressed_main(False, big["obj"], Y, SNPS, INDIV))      # bitwise repeatable
    finally:
        dg.set_engine(prev)
    lhs, rhs = (X * ZY).sum(dim=0), (ZtX * Y).sum(dim=0)
    assert float(((lhs - rhs).abs() / lhs.abs().clamp_min(1.0)).max()) <= 1e-10
    assert float(((ZY - ZY64).abs().amax(dim=0) / ZY64.abs().amax(dim=0)).max()) <= 1e-13
    rng = np.random.default_rng(2)
    ii = np.sort(rng.choice(INDIV, 32, replace=False))
    rows = big["plink_t"][torch.from_numpy(ii).to(big["dev"])].cpu().numpy()
    sub_plink = o.transpose_2bit(np.ascontiguousarray(rows), 32, SNPS)
    prob = dict(snps=SNPS, indiv=32, plink=sub_plink, plink_t=rows, f=np.zeros(SNPS))
    Bh = np.ascontiguousarray(Y.t().cpu().numpy())
    ref = o.dgemm_dense(0, prob, Bh, 0)                                                          # 32 x 32
    abssum = o.dgemm_dense(0, prob, np.abs(Bh), 0)
    got = ZY[torch.from_numpy(ii).to(big["dev"])].t().cpu().numpy()
    assert np.all(np.abs(got - ref) <= 3.02 * (S - 1) * 2.0 ** -53 * abssum)
