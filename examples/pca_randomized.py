#!/usr/bin/env python3
"""Randomised SNP principal components with the genotype matrix never decompressed: the PCA half of the reference's GBLUP example
(examples/gblup/calculate_gblup.jl:54-97 -- multiply_ld, randomized_range_finder, randomized_eigen, randomized_snp_pca; Halko et al. 2011).
Every multiply_ld is one 'N' and one 'T' dgemm_compressed with n = components + oversampling columns (50 by default) on device-resident
matrices; QR and the small eigenproblem run in torch on the same device.
usage: pca_randomized.py [--snps S] [--indiv N] [--components 10]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def multiply_ld(dg, obj, snps, indiv, B):
    """Zc^T (Zc B): SNPs x l  ->  SNPs x l   (calculate_gblup.jl:54-60)"""
    ZB = dg.dgemm_compressed_main(False, obj, B, snps, indiv)          # indiv x l
    return dg.dgemm_compressed_main(True, obj, ZB, snps, indiv)        # snps x l


def _colmajor(t):
    return t.t().contiguous().t()


def randomized_snp_pca(plink, plink_t, snps, indiv, freq, n, p=40, q=2, seed=0):
    """Returns (PC: indiv x n principal components of the centred genotype matrix, U: snps x n loadings), largest last like the reference."""
    import torch
    import miraculix_amd as mx
    dg = mx.dgemm_compressed
    dg.set_options(use_gpu=True, not_center=False, verbose=0)
    l = n + p
    obj = dg.init_compressed(plink, plink_t, snps, indiv, freq, l)
    try:
        g = torch.Generator(device=plink.device); g.manual_seed(seed)
        Omega = _colmajor(torch.randn((snps, l), dtype=torch.float64, device=plink.device, generator=g))
        Q = torch.linalg.qr(multiply_ld(dg, obj, snps, indiv, Omega)).Q
        for _ in range(q):                                              # two products per power iteration, as the reference does
            Q = torch.linalg.qr(multiply_ld(dg, obj, snps, indiv, _colmajor(Q))).Q
            Q = torch.linalg.qr(multiply_ld(dg, obj, snps, indiv, _colmajor(Q))).Q
        Q = _colmajor(Q)
        S = Q.t() @ multiply_ld(dg, obj, snps, indiv, Q)                # l x l, symmetric up to rounding
        ev, EV = torch.linalg.eigh(0.5 * (S + S.t()))                   # ascending: the largest components come last
        U = _colmajor(Q @ EV)
        PC = dg.dgemm_compressed_main(False, obj, U, snps, indiv)
        return PC[:, -n:], U[:, -n:], ev[-n:]
    finally:
        dg.free_compressed(obj)


def structured_population(torch, snps, indiv, seed, device, groups=3):
    """individuals from `groups` sub-populations with their own allele frequencies: the leading components separate them"""
    import numpy as np
    rng = np.random.default_rng(seed)
    base = rng.uniform(0.15, 0.5, size=snps)
    shift = rng.normal(0.0, 0.12, size=(groups, snps))
    lab = rng.integers(0, groups, size=indiv)
    P = np.clip(base[None, :] + shift[lab], 0.02, 0.98)
    Z = rng.binomial(2, P).astype(np.int8)                              # indiv x snps
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from _util import pack_plink
    plink = torch.from_numpy(np.ascontiguousarray(pack_plink(Z.T.copy()))).to(device)
    return plink, Z, lab


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--snps", type=int, default=20000)
    ap.add_argument("--indiv", type=int, default=2000)
    ap.add_argument("--components", type=int, default=10)
    args = ap.parse_args()
    import time
    import torch
    import miraculix_amd as mx
    mx.load_shared_library()
    dev = torch.device("cuda", 0)
    plink, Z, lab = structured_population(torch, args.snps, args.indiv, 1, dev)
    plink_t = mx.compressed_operations.transpose_genotype_matrix(plink, args.snps, args.indiv)
    f = mx.read_plink.calc_freq(plink, args.snps, args.indiv)
    t0 = time.perf_counter()
    PC, U, ev = randomized_snp_pca(plink, plink_t, args.snps, args.indiv, f, args.components)
    torch.cuda.synchronize()
    print(f"randomised PCA of {args.indiv} x {args.snps}: {args.components} components in {time.perf_counter() - t0:.3f} s; "
          f"largest eigenvalues of Zc^T Zc: {[f'{float(x):.4g}' for x in ev.flip(0)[:3]]}")


if __name__ == "__main__":
    main()
