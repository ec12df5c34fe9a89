#!/usr/bin/env python3
"""GBLUP on a small synthetic population, two ways that must agree (the two halves of the reference's examples/gblup and
examples/iterative_solver):
  dense     G = P Z Z^T P^T / (2 sum f(1-f)) from the int8 crossproduct (mxa_grm), then (G + lambda I) a = y by Cholesky
            (potrs_solve_gpu) -- the route of examples/gblup/calculate_gblup.jl
  iterative the same system by conjugate gradients with G never formed: G v = P Zc (Zc^T P^T v) / scale through mxa_gram_matvec
            -- the loop of examples/iterative_solver/grm_solve_cg.jl:74-134
usage: gblup_small.py [--snps S] [--indiv N] [--lam L]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(snps=20000, indiv=1500, lam=0.5, seed=3, verbose=True):
    import numpy as np
    import torch
    import miraculix_amd as mx
    from bench import synth_plink_device
    mx.load_shared_library()
    dev = torch.device("cuda", 0)
    plink = synth_plink_device(torch, snps, (indiv + 3) // 4, seed, dev)                  # SNP-major PLINK bytes
    plink_t = mx.compressed_operations.transpose_genotype_matrix(plink, snps, indiv)       # individual-major
    f = mx.read_plink.calc_freq(plink, snps, indiv)
    fh = f.cpu().numpy()
    scale = 2.0 * float((fh * (1.0 - fh)).sum())
    rng = np.random.default_rng(seed)
    y = rng.standard_normal((indiv, 1))

    # dense route
    G = mx.crossproduct.grm(plink_t, snps, indiv, is_plink_format=True, do_scale=True, allele_freq=fh).cpu().numpy()
    a_dense, logdet = mx.solve.dense_solve(G + lam * np.eye(indiv), y)

    # iterative route: G v = P Zc Zc^T P v / scale with P = I - 11^T/n (mxa_grm centres rows and columns of Z Z^T)
    dg = mx.dgemm_compressed
    dg.set_options(use_gpu=True, not_center=True, verbose=0)
    obj = dg.init_compressed(plink, plink_t, snps, indiv, f, 1)

    def Gv(v):
        v = v - v.mean()
        w = dg.gram_matvec(obj, v, snps, indiv)
        w = w - w.mean()
        return w / scale

    yd = torch.from_numpy(y).to(dev)
    x = torch.zeros_like(yd)
    r = yd - (Gv(x) + lam * x)
    p = r.clone()
    for it in range(1, 2000):
        rr = float((r * r).sum())
        if rr ** 0.5 < 1e-12 * float(torch.linalg.vector_norm(yd)):
            break
        Ap = Gv(p) + lam * p
        alpha = rr / float((p * Ap).sum())
        x = x + alpha * p
        r = r - alpha * Ap
        p = r + (float((r * r).sum()) / rr) * p
    dg.free_compressed(obj)
    a_cg = x.cpu().numpy()
    diff = float(np.abs(a_cg - a_dense).max() / np.abs(a_dense).max())
    if verbose:
        print(f"GBLUP {snps} SNPs x {indiv} indiv, lambda {lam}: Cholesky vs CG ({it} iterations) max rel diff {diff:.2e}, logdet(G + lambda I) = {logdet:.6f}")
    return diff, it


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--snps", type=int, default=20000)
    ap.add_argument("--indiv", type=int, default=1500)
    ap.add_argument("--lam", type=float, default=0.5)
    a = ap.parse_args()
    run(a.snps, a.indiv, a.lam)
