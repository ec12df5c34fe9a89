"""Config-5 loop shape (examples/iterative_solver/grm_solve_cg.jl:74-84,108-134) on device-resident vectors: the CG solve
through the n = 1 lookup kernel must agree with a dense fp64 solve of the same system."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples"))


def test_cg_matches_dense_solve():
    import torch
    import miraculix_amd as mx
    from miraculix_amd.distributed import HipLocalEngine, ShardedGenotypeOperator
    from grm_solve_cg import cg
    from _util import make_problem
    mx.load_shared_library()
    dev = torch.device("cuda", 0)
    snps, indiv = 3001, 517
    prob = make_problem(snps, indiv, 1, seed=12)
    eng = HipLocalEngine(torch.from_numpy(prob["plink"]).to(dev), torch.from_numpy(prob["plink_t"]).to(dev), snps, indiv, torch.from_numpy(prob["f"]).to(dev), 1, centered=True)
    op = ShardedGenotypeOperator(eng, snps, indiv)
    rng = np.random.default_rng(2)
    b = rng.standard_normal((indiv, 1))
    lam = float(snps)
    x, res, it = cg(op, torch.from_numpy(b).to(dev), torch.zeros((indiv, 1), dtype=torch.float64, device=dev), lam, max_iter=500, conv_crit=1e-10, verbose=False)
    eng.close()
    Zc = prob["Z"].astype(np.float64) - 2 * prob["f"][None, :]
    A = Zc @ Zc.T + lam * np.eye(indiv)
    x_ref = np.linalg.solve(A, b)
    assert res < 1e-9
    assert np.abs(x.cpu().numpy() - x_ref).max() <= 1e-9 * np.abs(x_ref).max()


def test_gram_matvec_matches_two_products_and_oracle():
    """mxa_gram_matvec = 'T' then 'N' with the intermediate on the device: bitwise equal to the two separate calls (same kernels,
    same order), and equal to the dense oracle within the stated tolerance; host and device operands, ld padding."""
    import torch
    import miraculix_amd as mx
    from _util import Oracle, make_B, make_problem
    mx.load_shared_library()
    dg = mx.dgemm_compressed
    o = Oracle()
    for snps, indiv, n, centered in [(3001, 517, 1, 1), (2050, 1301, 2, 1), (1500, 700, 5, 0), (4100, 515, 33, 1)]:
        prob = make_problem(snps, indiv, n, seed=snps)
        dg.set_options(use_gpu=True, not_center=not centered, verbose=0)
        obj = dg.init_compressed(prob["plink"], prob["plink_t"], snps, indiv, prob["f"], n)
        try:
            V = make_B(indiv, n, seed=9)                      # n x indiv
            Vcm = np.asfortranarray(V.T)
            G = dg.gram_matvec(obj, Vcm, snps, indiv)
            T = dg.dgemm_compressed_main(True, obj, Vcm, snps, indiv)
            N = dg.dgemm_compressed_main(False, obj, T, snps, indiv)
            assert np.array_equal(G, N)
            t_ref = o.dgemm_dense(1, prob, V, centered)[:, :snps]
            ref = o.dgemm_dense(0, prob, np.ascontiguousarray(t_ref), centered)[:, :indiv].T
            assert np.abs(G - ref).max() <= 1e-11 * np.abs(ref).max()
            dV = torch.from_numpy(V).cuda().t()                 # column-major on the device
            dOut = torch.full((n, indiv + 7), -1.0, dtype=torch.float64, device="cuda").t()[:indiv]   # ld = indiv + 7
            Gd = dg.gram_matvec(obj, dV, snps, indiv, out=dOut)
            assert np.array_equal(Gd.cpu().numpy(), G)
        finally:
            dg.free_compressed(obj)


def test_gblup_dense_route_equals_iterative_route():
    """examples/gblup_small.py: (G + lambda I) a = y through mxa_grm + potrs_solve_gpu equals the CG solve through mxa_gram_matvec --
    the crossproduct, the post-processing, the solver twin and the fp64 GEMM path tie up."""
    from gblup_small import run
    diff, it = run(snps=6000, indiv=700, lam=0.5, verbose=False)
    assert diff <= 1e-9 and it < 1000


def test_randomized_snp_pca_finds_the_population_structure():
    """examples/pca_randomized.py (the PCA half of the reference's examples/gblup/calculate_gblup.jl:54-97): n = 2 + 20 columns per
    multiply, two power iterations.  Three sub-populations give two dominant components; the randomised subspace must coincide with the
    one from a dense SVD of the centred matrix, and the eigenvalues with its squared singular values"""
    import torch
    import miraculix_amd as mx
    from pca_randomized import randomized_snp_pca, structured_population
    mx.load_shared_library()
    dev = torch.device("cuda", 0)
    snps, indiv, n = 3000, 400, 2
    plink, Z, lab = structured_population(torch, snps, indiv, 5, dev)
    plink_t = mx.compressed_operations.transpose_genotype_matrix(plink, snps, indiv)
    f = mx.read_plink.calc_freq(plink, snps, indiv)
    PC, U, ev = randomized_snp_pca(plink, plink_t, snps, indiv, f, n, p=20, q=2)
    Zc = Z.astype(np.float64) - 2.0 * f.cpu().numpy()[None, :]
    Ud, sd, Vt = np.linalg.svd(Zc, full_matrices=False)
    assert np.allclose(np.sort(ev.cpu().numpy())[::-1], sd[:n] ** 2, rtol=1e-8)
    # principal angles between the spans of the computed and the exact leading components
    Qa = np.linalg.qr(PC.cpu().numpy())[0]
    cosines = np.linalg.svd(Qa.T @ Ud[:, :n], compute_uv=False)
    assert cosines.min() >= 1.0 - 1e-9
    assert sd[n - 1] > 1.5 * sd[n]                                         # the structure is real: a gap after the second component


def test_gram_matvec_device_async_equals_sync():
    """mxa_gram_matvec_device with sync = 0: the step is only enqueued (on the object's blocking stream); torch ops issued right after it on
    the default stream see its result, and a chain of 20 dependent steps with torch arithmetic in between -- no host wait anywhere -- gives the
    same bits as the same chain through the synchronous entry.  Host pointers and multi-device objects are refused."""
    import ctypes
    import torch
    import miraculix_amd as mx
    from _util import make_problem
    L = mx.load_shared_library()
    dg = mx.dgemm_compressed
    dev = torch.device("cuda", 0)
    snps, indiv = 20_004, 3_001
    prob = make_problem(snps, indiv, 1, seed=21)
    dg.set_options(use_gpu=True, not_center=False, verbose=0)
    obj = dg.init_compressed(torch.from_numpy(prob["plink"]).to(dev), torch.from_numpy(prob["plink_t"]).to(dev), snps, indiv, torch.from_numpy(prob["f"]).to(dev), 1)
    try:
        g = torch.Generator(device=dev); g.manual_seed(5)
        v0 = torch.randn((1, indiv), dtype=torch.float64, device=dev, generator=g).t()

        def chain(sync):
            v = v0.clone()
            for _ in range(20):
                w = dg.gram_matvec(obj, v, snps, indiv, sync=sync)
                v = w / torch.linalg.vector_norm(w) + 0.25 * v          # torch ops on the default stream, dependent on the step
            torch.cuda.synchronize()
            return v
        a, b = chain(True), chain(False)
        assert torch.equal(a, b) and bool(torch.isfinite(a).all())
        hv = np.zeros((indiv, 1), order="F"); ho = np.zeros((indiv, 1), order="F")
        assert L.mxa_gram_matvec_device(obj, 1, hv.ctypes.data_as(ctypes.c_void_p), indiv, ho.ctypes.data_as(ctypes.c_void_p), indiv, 0) == 1
        assert L.mxa_last_error() != 0
    finally:
        dg.free_compressed(obj)
