#!/usr/bin/env python3
"""Conjugate-gradient solve of (G + lambda I) x = b with G = Zc Zc^T never formed: every iteration is one 'T' and one 'N'
dgemm_compressed with n = 1 on device-resident vectors.  Mirrors the loop of the reference's
examples/iterative_solver/grm_solve_cg.jl:74-84 (GRM_vec) and :108-134 (CG), with all vectors kept in HBM and the SNPs
optionally sharded over the ranks of a torch.distributed job (one fp64 all-reduce of `indiv` doubles per iteration).

single GPU :  python examples/grm_solve_cg.py --snps 200000 --indiv 20000
multi GPU  :  python -m torch.distributed.run --nproc-per-node N examples/grm_solve_cg.py ...
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def grm_vec(op, v, lam):
    """(Zc Zc^T + lam I) v : 'T' on this rank's SNP block, then 'N', fused into one library call (mxa_gram_matvec) with the
    snps_local x 1 intermediate kept in HBM; summed over the ranks inside op.gram"""
    return op.gram(v) + lam * v


def cg(op, b, x0, lam, max_iter=1000, conv_crit=1e-2, print_iter=100, verbose=True):
    import torch
    x = x0.clone()
    r = b - grm_vec(op, x, lam)
    p = r.clone()
    it = 0
    for it in range(1, max_iter + 1):
        norm_old = torch.linalg.vector_norm(r)
        if float(norm_old) < conv_crit:
            break
        gp = grm_vec(op, p, lam)
        alpha = norm_old**2 / (p * gp).sum()
        if verbose and it % print_iter == 0:
            print(float(alpha), float(norm_old), flush=True)
        x = x + alpha * p
        r = r - alpha * gp
        beta = torch.linalg.vector_norm(r) ** 2 / norm_old**2
        p = r + beta * p
    return x, float(torch.linalg.vector_norm(r)), it


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--snps", type=int, default=200_000)
    ap.add_argument("--indiv", type=int, default=20_000)
    ap.add_argument("--lam", type=float, default=None, help="ridge term; default = snps (well-conditioned toy system)")
    ap.add_argument("--max-iter", type=int, default=200)
    ap.add_argument("--engine", choices=["f64", "i8", "f64-strict", "i8-exact"], default="f64",
                    help="f64 (default): at n = 1 the exact radix-256 splitting of the vector on the int8 matrix cores whenever a per-call check proves it exact "
                         "(HBM-bound), else fp64 pair tables; f64-strict: fp64 arithmetic only; i8: the splitting without the check")
    args = ap.parse_args()
    import torch
    import torch.distributed as dist
    from bench import synth_plink_device
    import miraculix_amd as mx
    from miraculix_amd.distributed import HipLocalEngine, ShardedGenotypeOperator, shard_bounds
    world = int(os.environ.get("WORLD_SIZE", "1")); rank = int(os.environ.get("RANK", "0")); lr = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(lr)
    dev = torch.device("cuda", lr)
    os.environ["HIP_DEVICE"] = str(lr)
    if world > 1:
        dist.init_process_group("nccl", device_id=dev)
    L = mx.load_shared_library()
    mx.dgemm_compressed.set_engine(args.engine)
    b0, e0 = shard_bounds(args.snps, world, rank)
    sl = e0 - b0
    plink = synth_plink_device(torch, sl, (args.indiv + 3) // 4, 42 + rank, dev)
    plink_t = mx.compressed_operations.transpose_genotype_matrix(plink, sl, args.indiv)
    f = mx.read_plink.calc_freq(plink, sl, args.indiv)
    eng = HipLocalEngine(plink, plink_t, sl, args.indiv, f, 1, centered=True)
    del plink, plink_t
    op = ShardedGenotypeOperator(eng, args.snps, args.indiv)
    g = torch.Generator(device=dev); g.manual_seed(7)
    b = torch.randn((args.indiv, 1), dtype=torch.float64, device=dev, generator=g)
    x0 = torch.zeros_like(b)
    lam = args.lam if args.lam is not None else float(args.snps)
    cg(op, b, x0, lam, max_iter=2, conv_crit=0.0, verbose=False)   # warm-up: workspace growth, first launches and torch's lazily loaded kernels stay out of the timing
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    x, res, it = cg(op, b, x0, lam, max_iter=args.max_iter, conv_crit=1e-8 * float(torch.linalg.vector_norm(b)), verbose=(rank == 0))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if rank == 0:
        flops = 2 * 2.0 * args.snps * args.indiv * (it + 1)
        print(f"CG: {it} iterations, residual {res:.3e}, {dt*1e3:.1f} ms total, {dt/(it+1)*1e3:.3f} ms per G*v, {flops/dt*1e-12:.2f} TFLOP/s effective (n=1, engine {args.engine})")
    eng.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
