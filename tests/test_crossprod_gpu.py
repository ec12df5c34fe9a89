"""GPU parity for the integer crossproduct (snp_multiply_gpu): bit-exact against the int32 oracle.
Mirrors tests/crossproduct/test_grm.jl:114-157 of the reference (random 0/1/2 matrices, uneven dimensions,
exact-integer comparison against a dense GEMM) and test_ld.jl."""
import numpy as np
import pytest

from _util import Oracle, make_problem, pack_plink

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mx():
    import miraculix_amd as m
    m.load_shared_library()
    return m


@pytest.mark.parametrize("n_snps,n_indiv", [(953, 752), (10251, 752), (953, 1343), (5000, 300), (131, 17), (4096, 512), (20000, 2100)])
def test_crossprod_plink_exact(mx, n_snps, n_indiv):
    o = Oracle()
    prob = make_problem(n_snps, n_indiv, 1, seed=n_snps + n_indiv)
    X = prob["plink_t"]                                     # indiv rows x ceil(snps/4)
    M = mx.crossproduct.snp_crossprod(X, n_snps, n_indiv, is_snpmajor=False, is_plink_format=True)
    ref = o.crossprod_i32(X, n_snps, True)
    assert M.shape == (n_indiv, n_indiv)
    assert np.array_equal(M, M.T)
    assert np.array_equal(M, ref.astype(np.float64))
    # analytic oracle of the reference test: BLAS gemm on the decoded 0/1/2 matrix
    Z = prob["Z"].astype(np.float64)
    assert np.array_equal(M, Z @ Z.T)


def test_crossprod_raw_2bit_and_snpmajor(mx):
    o = Oracle()
    rng = np.random.default_rng(3)
    rows, k = 333, 1777
    V = rng.integers(0, 4, size=(rows, k)).astype(np.uint8)   # raw 2-bit values 0..3 (is_plink_format = false)
    pad = (-k) % 4
    Vp = np.concatenate([V, np.zeros((rows, pad), np.uint8)], axis=1).reshape(rows, -1, 4)
    X = (Vp[:, :, 0] | (Vp[:, :, 1] << 2) | (Vp[:, :, 2] << 4) | (Vp[:, :, 3] << 6)).astype(np.uint8)
    M = mx.crossproduct.snp_crossprod(X, rows, k, is_snpmajor=True, is_plink_format=False)  # "snps" rows -> snps x snps
    ref = o.crossprod_i32(X, k, False)
    assert np.array_equal(M, ref.astype(np.float64))
    assert np.array_equal(M, V.astype(np.float64) @ V.astype(np.float64).T)


def test_crossprod_missing_byte_quirk(mx):
    """a byte holding a missing pair (01) reads as 0xFF = four 3s, exactly like the reference's table (snp_multiply_cuda.h:202)"""
    o = Oracle()
    prob = make_problem(801, 130, 1, seed=11, missing_frac=0.02)
    X = prob["plink_t"]
    M = mx.crossproduct.snp_crossprod(X, 801, 130, is_snpmajor=False, is_plink_format=True)
    ref = o.crossprod_i32(X, 801, True)
    assert np.array_equal(M, ref.astype(np.float64))


def test_grm_and_ld(mx):
    prob = make_problem(3000, 400, 1, seed=21)
    Z = prob["Z"].astype(np.float64)
    f = prob["f"]
    G = mx.crossproduct.grm(prob["plink_t"], 3000, 400, is_plink_format=True, do_scale=True, allele_freq=f)
    Zc = Z - Z.mean(axis=0, keepdims=True)
    Gref = Zc @ Zc.T / (2 * np.sum(f * (1 - f)))
    assert np.abs(G - Gref).max() <= 1e-9 * np.abs(Gref).max()
    R = mx.crossproduct.ld(prob["plink"], 3000, 400, is_plink_format=True, allele_freq=f)
    Mld = Z.T @ Z - 4 * 400 * np.outer(f, f)
    s = np.sqrt(np.diag(Mld))
    assert np.abs(R - Mld / s[:, None] / s[None, :]).max() <= 1e-9


def test_grm_device_resident(mx):
    """mxa_grm with every operand in HBM (torch tensors): crossproduct + rank-1 centring + scaling never leave the device"""
    import torch
    dev = torch.device("cuda", 0)
    prob = make_problem(2500, 333, 1, seed=4)
    f = prob["f"]
    G = mx.crossproduct.grm(torch.from_numpy(prob["plink_t"]).to(dev), 2500, 333, is_plink_format=True, do_scale=True, allele_freq=torch.from_numpy(f).to(dev))
    assert G.is_cuda
    Z = prob["Z"].astype(np.float64)
    Zc = Z - Z.mean(axis=0, keepdims=True)
    Gref = Zc @ Zc.T / (2 * np.sum(f * (1 - f)))
    assert np.abs(G.cpu().numpy() - Gref).max() <= 1e-9 * np.abs(Gref).max()
    G2 = mx.crossproduct.grm(prob["plink_t"], 2500, 333, is_plink_format=True, do_scale=False)
    assert np.abs(G2 - Zc @ Zc.T).max() <= 1e-9 * np.abs(Zc @ Zc.T).max()


@pytest.mark.parametrize("rows,k", [(1000, 900), (1537, 333)])
def test_crossprod_panels_tile_the_full_result(rows, k):
    """mxa_snp_multiply_panel (output-tile sharding): panels are bit-identical slabs of snp_multiply_gpu's result;
    upper_only computes rows [0, col_end) and leaves zeros below; host and device operands."""
    import torch
    import miraculix_amd as mx
    from miraculix_amd.distributed import ShardedCrossproduct, panel_bounds
    mx.load_shared_library()
    rng = np.random.default_rng(rows)
    Z = rng.integers(0, 3, size=(rows, k)).astype(np.int8)
    from _util import pack_plink
    X = np.ascontiguousarray(pack_plink(Z))                       # PLINK codes, rows x ceil(k/4)
    full = mx.crossproduct.snp_crossprod(X, k, rows, is_snpmajor=False, is_plink_format=True)
    ref = Z.astype(np.int64) @ Z.astype(np.int64).T
    assert np.array_equal(full, ref.astype(np.float64))
    for world in (1, 2, 3):
        for rank in range(world):
            for balance in (False, True):
                c0, c1 = panel_bounds(rows, world, rank, balance)
                if c1 <= c0:
                    continue
                P = mx.crossproduct.snp_crossprod_panel(X, k, rows, c0, c1, upper_only=balance, is_plink_format=True)
                expect = full[c0:c1, :].copy()
                if balance:
                    expect[:, c1:] = 0.0
                assert np.array_equal(P, expect), (world, rank, balance)
    dX = torch.from_numpy(X).cuda()
    c0, c1, P = ShardedCrossproduct(lambda a, b, u: mx.crossproduct.snp_crossprod_panel(dX, k, rows, a, b, upper_only=u, is_plink_format=True), rows).compute()
    assert (c0, c1) == (0, rows) and np.array_equal(P.cpu().numpy(), full)
    with pytest.raises(RuntimeError):
        mx.crossproduct.snp_crossprod_panel(X, k, rows, 100, 512, is_plink_format=True)   # col_begin must be a multiple of 256


def test_crossprod_host_result_is_pipelined_in_slabs(monkeypatch):
    """Host result: tile rows are launched in chunks and each finished column slab is copied out while the next chunk computes.
    Small slabs force many chunks; the result must be the same matrix, and identical to the unpipelined path."""
    import miraculix_amd as mx
    from _util import pack_plink
    mx.load_shared_library()
    rng = np.random.default_rng(5)
    rows, k = 2305, 700
    Z = rng.integers(0, 3, size=(rows, k)).astype(np.int8)
    X = np.ascontiguousarray(pack_plink(Z))
    ref = (Z.astype(np.int64) @ Z.astype(np.int64).T).astype(np.float64)
    monkeypatch.setenv("MXA_XPROD_SLAB_MB", "5")          # 5 MiB -> one tile row (256 columns) per chunk, 10 chunks
    M = mx.crossproduct.snp_crossprod(X, k, rows, is_snpmajor=False, is_plink_format=True)
    assert np.array_equal(M, ref)
    monkeypatch.setenv("MXA_XPROD_NO_PIPELINE", "1")
    M2 = mx.crossproduct.snp_crossprod(X, k, rows, is_snpmajor=False, is_plink_format=True)
    assert np.array_equal(M2, ref)


@pytest.mark.parametrize("slab_mb,rows", [("5", 2305), ("9", 2305), ("3", 1290)])
def test_crossprod_host_result_through_the_slab_ring(monkeypatch, slab_mb, rows):
    """Host result without an n x n device buffer (round 4: crossprod_to_host_ring): the matrix is produced column slab by column slab into a ring
    of three device buffers -- every slab the column panel of mxa_snp_multiply_panel -- and copied out while the next slab computes.  Forced here
    (MXA_XPROD_HOST_RING=2) with small slabs so that the ring wraps several times (10 slabs of one tile column, 5 of two, 6 of one); the default
    takes it for results of 4 GB and more whose download outlasts twice the triangular arithmetic (tests/test_fullsize_configs_gpu.py runs
    that at 12.8 GB).  Bit-exact against the integer product, for the plain crossproduct, GRM and LD (fused post-processing in the epilogue)."""
    import miraculix_amd as mx
    from _util import pack_plink
    mx.load_shared_library()
    rng = np.random.default_rng(6)
    k = 900
    Z = rng.integers(0, 3, size=(rows, k)).astype(np.int8)
    X = np.ascontiguousarray(pack_plink(Z))
    ref = (Z.astype(np.int64) @ Z.astype(np.int64).T).astype(np.float64)
    monkeypatch.setenv("MXA_XPROD_SLAB_MB", slab_mb)
    monkeypatch.setenv("MXA_XPROD_HOST_RING", "0")
    M0 = mx.crossproduct.snp_crossprod(X, k, rows, is_snpmajor=False, is_plink_format=True)
    f = Z.astype(np.float64).mean(axis=0) / 2.0
    G0 = mx.crossproduct.grm(X, k, rows, is_plink_format=True, do_scale=True, allele_freq=f)
    fr = Z.astype(np.float64).mean(axis=1) / 2.0
    R0 = mx.crossproduct.ld(X, rows, k, is_plink_format=True, allele_freq=fr)
    monkeypatch.setenv("MXA_XPROD_HOST_RING", "2")
    M = np.full((rows, rows), -5.0)
    mx.crossproduct.snp_crossprod(X, k, rows, is_snpmajor=False, is_plink_format=True, out=M)
    assert np.array_equal(M, ref) and np.array_equal(M0, ref)
    G = mx.crossproduct.grm(X, k, rows, is_plink_format=True, do_scale=True, allele_freq=f)
    assert np.array_equal(G, G0)
    R = mx.crossproduct.ld(X, rows, k, is_plink_format=True, allele_freq=fr)
    assert np.array_equal(R, R0, equal_nan=True)
