"""Parity at the full single-GPU sizes of BASELINE.json configs 3, 4 and 5 (config 2 is tests/test_fullsize_gpu.py), where the
oracle cannot run on the whole problem: size-independent properties plus sampled rows / tiles against the oracle on the
extracted sub-problem.  Shapes follow the reference's tests/crossproduct/test_grm.jl:114-157 (crossproduct against a dense
product on 0/1/2, symmetry) and examples/iterative_solver/grm_solve_cg.jl:74-84 (the 'T' + 'N' pair of a CG step).

  C3  GRM crossproduct 500 000 SNPs x 100 000 individuals into a device-resident 80 GB result (k_xstage over a 12.5 GB source,
      76 k tiles, column offsets beyond 2^32 elements)
  C4  per-GPU shard of 5M x 200k over 8 GPUs: 625 000 SNPs x 200 000 individuals, ncol = 128, allele-frequency centred
  C5  per-GPU shard of 2M x 100k over 8 GPUs: 250 000 SNPs x 100 000 individuals, n = 1 (one CG step)
"""
import numpy as np
import pytest

from _util import Oracle

pytestmark = pytest.mark.gpu

RTOL = 1e-11   # stated fp64 tolerance of the path: max|C - C_ref| <= 1e-11 max|C_ref|


def _mods():
    import torch
    import miraculix_amd as mx
    mx.load_shared_library()
    torch.cuda.empty_cache()
    return torch, mx, torch.device("cuda", 0)


def _stage(torch, mx, dev, snps, indiv, n, seed):
    from bench import synth_genotypes_device
    plink = synth_genotypes_device(torch, snps, indiv, seed, dev)                       # SNP-major, Binomial(2, p_s)
    plink_t = mx.compressed_operations.transpose_genotype_matrix(plink, snps, indiv)
    f = mx.read_plink.calc_freq(plink, snps, indiv)
    dg = mx.dgemm_compressed
    dg.set_options(use_gpu=True, not_center=False, verbose=0)
    obj = dg.init_compressed(plink, plink_t, snps, indiv, f, n)
    return dict(torch=torch, mx=mx, dev=dev, plink=plink, plink_t=plink_t, f=f, obj=obj, dg=dg, snps=snps, indiv=indiv, n=n)


def _sampled_vs_oracle(S, trans, Bdev, Cdev, cols, centered, nsample=64, seed=1):
    """nsample rows of the result against the long-double dense oracle on the extracted rows of the packed matrix (the helper bench.py's
    config legs use too)"""
    from bench import sampled_rows_vs_oracle
    err = sampled_rows_vs_oracle(S["torch"], S, trans, Bdev, Cdev, cols, centered, nsample=nsample, seed=seed)
    assert err <= RTOL, (trans, err)


# ====================================================================================================== config 4 shard
@pytest.fixture(scope="module")
def c4():
    torch, mx, dev = _mods()
    S = _stage(torch, mx, dev, 625_000, 200_000, 128, seed=44)
    yield S
    S["dg"].free_compressed(S["obj"])
    S.clear()
    torch.cuda.empty_cache()


def test_c4_checksums_exact_n128(c4):
    """uncentred, B = ones in all 128 columns: every column of 'T' is the per-SNP allele count (= 2 indiv f_s from the independent
    popcount kernel), every column of 'N' the per-individual allele count -- exact integers, all four column chunks identical"""
    torch, mx, dg, dev = c4["torch"], c4["mx"], c4["dg"], c4["dev"]
    snps, indiv, n = c4["snps"], c4["indiv"], c4["n"]
    dg.set_options(use_gpu=True, not_center=True, verbose=0)      # centring is a process-global option read at call time
    try:
        ones_i = torch.ones((n, indiv), dtype=torch.float64, device=dev).t()
        ct = dg.dgemm_compressed_main(True, c4["obj"], ones_i, snps, indiv)            # snps x 128
        counts_s = torch.round(c4["f"] * (2.0 * indiv))
        assert torch.equal(ct, counts_s[:, None].expand(snps, n))
        del ct
        ones_s = torch.ones((n, snps), dtype=torch.float64, device=dev).t()
        cn = dg.dgemm_compressed_main(False, c4["obj"], ones_s, snps, indiv)           # indiv x 128
        fi = mx.read_plink.calc_freq(c4["plink_t"], indiv, snps)                        # the transposed matrix read as "indiv SNPs"
        assert torch.equal(cn, torch.round(fi * (2.0 * snps))[:, None].expand(indiv, n))
        assert float(cn[:, 5].sum()) == float(counts_s.sum())
    finally:
        dg.set_options(use_gpu=True, not_center=False, verbose=0)


def test_c4_centred_adjoint_repeatable_and_sampled(c4):
    torch, dg, dev = c4["torch"], c4["dg"], c4["dev"]
    snps, indiv, n = c4["snps"], c4["indiv"], c4["n"]
    g = torch.Generator(device=dev); g.manual_seed(3)
    Y = torch.randn((n, snps), dtype=torch.float64, device=dev, generator=g).t()       # snps x n
    X = torch.randn((n, indiv), dtype=torch.float64, device=dev, generator=g).t()      # indiv x n
    ZY = dg.dgemm_compressed_main(False, c4["obj"], Y, snps, indiv)                     # indiv x n, centred
    ZtX = dg.dgemm_compressed_main(True, c4["obj"], X, snps, indiv)                     # snps x n, centred
    # adjoint identity of the CENTRED operator ties the two stored orientations and both centring epilogues together
    lhs = (X * ZY).sum(dim=0)
    rhs = (ZtX * Y).sum(dim=0)
    scale = (X.abs() * ZY.abs()).sum(dim=0)
    assert float(((lhs - rhs).abs() / scale).max()) <= 1e-11
    # bitwise reproducible (fixed split-K order, no atomics)
    assert torch.equal(ZY, dg.dgemm_compressed_main(False, c4["obj"], Y, snps, indiv))
    assert torch.equal(ZtX, dg.dgemm_compressed_main(True, c4["obj"], X, snps, indiv))
    cols = [0, 1, 31, 32, 63, 64, 126, 127]                                            # both ends of the four 32-column chunks
    _sampled_vs_oracle(c4, 0, Y, ZY, cols, centered=1)
    _sampled_vs_oracle(c4, 1, X, ZtX, cols, centered=1)


# ====================================================================================================== config 5 shard
@pytest.fixture(scope="module")
def c5():
    torch, mx, dev = _mods()
    S = _stage(torch, mx, dev, 250_000, 100_000, 1, seed=45)
    yield S
    S["dg"].free_compressed(S["obj"])
    S.clear()
    torch.cuda.empty_cache()


def test_c5_cg_step_n1(c5):
    """one CG step G v = Zc (Zc^T v), n = 1: mxa_gram_matvec is bitwise the 'T' call followed by the 'N' call, both products
    agree with the oracle on sampled rows, and the step is bitwise repeatable"""
    torch, dg, dev = c5["torch"], c5["dg"], c5["dev"]
    snps, indiv = c5["snps"], c5["indiv"]
    g = torch.Generator(device=dev); g.manual_seed(7)
    v = torch.randn((1, indiv), dtype=torch.float64, device=dev, generator=g).t()       # indiv x 1
    T = dg.dgemm_compressed_main(True, c5["obj"], v, snps, indiv)                        # snps x 1
    N = dg.dgemm_compressed_main(False, c5["obj"], T, snps, indiv)                       # indiv x 1
    G = dg.gram_matvec(c5["obj"], v, snps, indiv)
    assert torch.equal(G, N)
    assert torch.equal(G, dg.gram_matvec(c5["obj"], v, snps, indiv))
    _sampled_vs_oracle(c5, 1, v, T, [0], centered=1)
    _sampled_vs_oracle(c5, 0, T, N, [0], centered=1)
    # v^T G v = |Zc^T v|^2 > 0
    q = float((v * G).sum()); t2 = float((T * T).sum())
    assert abs(q - t2) <= 1e-11 * t2
    # n = 2 takes the same small-n path with two columns: column 0 must not depend on its neighbour
    v2 = torch.cat([v, -2.0 * v], dim=1).t().contiguous().t()
    T2 = dg.dgemm_compressed_main(True, c5["obj"], v2, snps, indiv)
    assert float((T2[:, :1] - T).abs().max()) <= RTOL * float(T.abs().max())
    assert float((T2[:, 1:] + 2.0 * T).abs().max()) <= 2 * RTOL * float(T.abs().max())


# ====================================================================================================== config 3
@pytest.fixture(scope="module")
def c3():
    torch, mx, dev = _mods()
    from bench import synth_genotypes_device
    snps, indiv = 500_000, 100_000
    X = synth_genotypes_device(torch, indiv, snps, 46, dev, p_along="cols")              # individual-major, 12.5 GB
    M = torch.empty((indiv, indiv), dtype=torch.float64, device=dev)                     # 80 GB
    M.fill_(-1.0)
    mx.crossproduct.snp_crossprod(X, snps, indiv, is_snpmajor=False, is_plink_format=True, out=M)
    yield dict(torch=torch, mx=mx, dev=dev, X=X, M=M, snps=snps, indiv=indiv)
    del X, M
    torch.cuda.empty_cache()


def test_c3_symmetry_and_diagonal(c3):
    torch, M, X, n = c3["torch"], c3["M"], c3["X"], c3["indiv"]
    # exact symmetry, checked panel by panel (no second 80 GB matrix)
    step = 2048
    for a in range(0, n, step):
        b = min(n, a + step)
        assert torch.equal(M[a:b, :], M[:, a:b].t()), (a, b)
    # diagonal = sum_s z^2 = #(code 10) + 4 #(code 11), from an independent bit-count pass over the raw bytes
    pop = torch.tensor([bin(i).count("1") for i in range(256)], dtype=torch.int32, device=c3["dev"])
    diag = torch.zeros(n, dtype=torch.float64, device=c3["dev"])
    rows_per = max(1, (128 << 20) // X.shape[1])
    for r0 in range(0, n, rows_per):
        b = X[r0:r0 + rows_per]
        H, L = (b >> 1) & 0x55, b & 0x55
        ones, twos = H & ~L & 0x55, H & L
        diag[r0:r0 + rows_per] = (pop[ones.long()] + 4 * pop[twos.long()]).sum(dim=1).to(torch.float64)
    assert torch.equal(torch.diagonal(M), diag)
    assert float(M.min()) >= 0.0                                                         # every entry was written (fill value -1)


def test_c3_sampled_tiles_vs_oracle(c3):
    """eight 256 x 256 tiles against the exact integer oracle on the extracted rows: the first and the last (ragged, 160 rows)
    diagonal tiles, tiles in the far corner whose element offsets i*n + j exceed 2^32, and random interior tiles"""
    torch, M, X, n, snps = c3["torch"], c3["M"], c3["X"], c3["indiv"], c3["snps"]
    o = Oracle()
    nb = (n + 255) // 256
    rng = np.random.default_rng(5)
    tiles = [(0, 0), (nb - 1, nb - 1), (0, nb - 1), (nb - 2, nb - 1), (200, 300)]
    while len(tiles) < 8:
        i, j = sorted(rng.integers(0, nb, 2).tolist())
        tiles.append((int(i), int(j)))
    assert any(i * 256 * n + j * 256 > 2 ** 32 for i, j in tiles)
    for ti, tj in tiles:
        ri = np.arange(ti * 256, min(n, ti * 256 + 256)); rj = np.arange(tj * 256, min(n, tj * 256 + 256))
        rows = np.concatenate([ri, rj]) if ti != tj else ri
        sub = np.ascontiguousarray(X[torch.from_numpy(rows).to(c3["dev"])].cpu().numpy())
        ref = o.crossprod_i32(sub, snps, True).astype(np.float64)
        if ti != tj:
            ref = ref[: len(ri), len(ri):]
        got = M[ri[0]:ri[-1] + 1, rj[0]:rj[-1] + 1].cpu().numpy()
        assert np.array_equal(got, ref), (ti, tj)
        got_m = M[rj[0]:rj[-1] + 1, ri[0]:ri[-1] + 1].cpu().numpy()
        assert np.array_equal(got_m, ref.T), (ti, tj)


def test_c3_host_result_through_the_plain_abi():
    """the reference's own calling convention (host snp_matrix, host ans; crossproduct.jl:54-58) at K = 500 000 with a 12.8 GB result: the
    staged upload, the tile-row chunks and the four copier threads of the pipelined copy-out (mxa_crossprod.hip: crossprod_to_host) against the
    device-resident result of the same rows (bitwise) and the oracle on a sampled tile"""
    import torch
    import miraculix_amd as mx
    from bench import synth_genotypes_device
    mx.load_shared_library()
    torch.cuda.empty_cache()
    dev = torch.device("cuda", 0)
    snps, rows = 500_000, 40_000
    Xd = synth_genotypes_device(torch, rows, snps, 47, dev, p_along="cols")
    X = Xd.cpu().numpy()                                                     # 5 GB host copy
    M = mx.crossproduct.snp_crossprod(X, snps, rows, is_snpmajor=False, is_plink_format=True)      # host in, host out
    assert M.shape == (rows, rows)
    Md = mx.crossproduct.snp_crossprod(Xd, snps, rows, is_snpmajor=False, is_plink_format=True)    # device in, device out
    step = 4000
    for a in range(0, rows, step):
        assert np.array_equal(M[a:a + step], Md[a:a + step].cpu().numpy()), a
    del Md
    o = Oracle()
    ri, rj = np.arange(256, 512), np.arange(rows - 160, rows)
    sub = np.ascontiguousarray(X[np.concatenate([ri, rj])])
    ref = o.crossprod_i32(sub, snps, True).astype(np.float64)[:256, 256:]
    assert np.array_equal(M[256:512, rows - 160:], ref)
    del Xd
    torch.cuda.empty_cache()


def test_c5_cg_loop_on_the_full_shard(c5):
    """the loop of config 5 (examples/iterative_solver/grm_solve_cg.jl:108-134) on the full per-GPU shard: 25 CG iterations of
    (Zc Zc^T + lambda I) x = b with one mxa_gram_matvec each.  The residual must fall monotonically for this well-conditioned system, the final x
    must satisfy the equation to the residual the loop reports (checked with two separate products), and a second run must repeat bit for bit"""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples"))
    from grm_solve_cg import cg
    torch, dg, dev = c5["torch"], c5["dg"], c5["dev"]
    snps, indiv, obj = c5["snps"], c5["indiv"], c5["obj"]

    class Op:   # the part of ShardedGenotypeOperator the loop uses, on the fixture's object
        def gram(self, v):
            return dg.gram_matvec(obj, v, snps, indiv)

    g = torch.Generator(device=dev); g.manual_seed(11)
    b = torch.randn((1, indiv), dtype=torch.float64, device=dev, generator=g).t()
    lam = float(snps)
    x0 = torch.zeros_like(b)
    hist = []

    class Rec(Op):
        def gram(self, v):
            out = super().gram(v)
            hist.append(float(torch.linalg.vector_norm(out)))
            return out

    x, res, it = cg(Rec(), b, x0, lam, max_iter=25, conv_crit=1e-30, verbose=False)
    assert it == 25 and len(hist) == 26
    T = dg.dgemm_compressed_main(True, obj, x, snps, indiv)
    Ax = dg.dgemm_compressed_main(False, obj, T, snps, indiv) + lam * x
    true_res = float(torch.linalg.vector_norm(b - Ax))
    assert res < 1e-6 * float(torch.linalg.vector_norm(b))                 # well-conditioned: 25 iterations go far
    assert abs(true_res - res) <= 1e-6 * float(torch.linalg.vector_norm(b)) + 10 * res
    x2, res2, _ = cg(Op(), b, x0, lam, max_iter=25, conv_crit=1e-30, verbose=False)
    assert torch.equal(x, x2) and res == res2
