"""N > 1 path on CPU: world_size-2 gloo run of the SNP-sharded operator (miraculix_amd/distributed.py) with the oracle as the
local engine (test-only), checked against the unsharded dense oracle.  Covers the partition, the 'N' sum all-reduce with
the centring partial sums riding along, and the collective-free 'T' row blocks."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from miraculix_amd.distributed import ShardedGenotypeOperator, shard_bounds


def test_shard_bounds_cover_and_align():
    for snps in [1, 4, 1000, 1003, 999_999]:
        for world in [1, 2, 3, 8]:
            edges = [shard_bounds(snps, world, r) for r in range(world)]
            assert edges[0][0] == 0 and edges[-1][1] == snps
            for (b0, e0), (b1, e1) in zip(edges, edges[1:]):
                assert e0 == b1
            assert all(b % 4 == 0 or b == e for b, e in edges)   # non-empty shards start on a packed-byte boundary


class OracleLocalEngine:
    """TEST-ONLY local engine (the product's engine is HipLocalEngine): SNP block [b, e) through the dense CPU oracle"""

    def __init__(self, prob, b, e, centered):
        from _util import Oracle
        self.o = Oracle()
        Z = prob["Z"][:, b:e]
        from _util import pack_plink
        self.prob = dict(snps=e - b, indiv=prob["indiv"], plink=np.ascontiguousarray(pack_plink(Z.T.copy())), plink_t=np.ascontiguousarray(pack_plink(Z)), f=np.ascontiguousarray(prob["f"][b:e]))
        self.centered = centered

    def multiply(self, transpose, B, out=None):
        Bn = np.ascontiguousarray(B.numpy().T)  # (n x k) rows = columns
        C = self.o.dgemm_dense(int(transpose), self.prob, Bn, self.centered)
        return torch.from_numpy(np.ascontiguousarray(C)).t()  # (m x n) column-major view


def _worker(rank, world, port, snps, indiv, n, centered, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from _util import Oracle, make_B, make_problem
    prob = make_problem(snps, indiv, n, seed=5)
    b, e = shard_bounds(snps, world, rank)
    op = ShardedGenotypeOperator(OracleLocalEngine(prob, b, e, centered), snps, indiv)
    BN = make_B(snps, n, seed=1)   # (n x snps)
    BT = make_B(indiv, n, seed=2)
    CN = op.matmul_N(torch.from_numpy(np.ascontiguousarray(BN[:, b:e])).t())
    CT = op.matmul_T(torch.from_numpy(BT).t())
    o = Oracle()
    refN = o.dgemm_dense(0, prob, BN, centered)
    refT = o.dgemm_dense(1, prob, BT, centered)
    errN = np.abs(CN.t().numpy() - refN).max() / np.abs(refN).max()
    errT = np.abs(CT.t().numpy() - refT[:, b:e]).max() / np.abs(refT).max()
    # G V = Zc Zc^T V: every rank applies its SNP block, one all-reduce (the CG step of examples/grm_solve_cg.py)
    GV = op.gram(torch.from_numpy(BT).t())
    refG = o.dgemm_dense(0, prob, np.ascontiguousarray(refT[:, :snps]), centered)
    errG = np.abs(GV.t().numpy() - refG).max() / np.abs(refG).max()
    ret[rank] = (float(errN), float(errT), float(errG))
    dist.destroy_process_group()


@pytest.mark.parametrize("centered", [0, 1])
def test_world2_gloo_sharded_operator(centered):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(2, port, 1003, 301, 5, centered, ret), nprocs=2, join=True)
    assert len(ret) == 2
    for r in range(2):
        errN, errT, errG = ret[r]
        assert errN <= 1e-12 and errT <= 1e-12 and errG <= 1e-12


def _xworker(rank, world, port, n, k, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from miraculix_amd.distributed import ShardedCrossproduct, XTILE
    rng = np.random.default_rng(3)
    X = rng.integers(0, 3, size=(n, k)).astype(np.int64)
    M = (X @ X.T).astype(np.float64)

    def panel_fn(c0, c1, upper_only):   # TEST-ONLY local engine with the semantics of mxa_snp_multiply_panel
        P = M[c0:c1, :].copy()          # P[c, r] = M[r, c0 + c] (M symmetric)
        if upper_only:
            P[:, c1:] = 0.0             # rows below the panel's diagonal block are not computed
        return torch.from_numpy(P)

    res = {}
    for exch in (False, True):
        c0, c1, P = ShardedCrossproduct(panel_fn, n).compute(exchange_symmetric=exch)
        res[exch] = (c0, c1, None if P is None else bool(np.array_equal(P.numpy(), M[c0:c1, :])))
    ret[rank] = res
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n", [(2, 1500), (3, 2600)])
def test_gloo_sharded_crossproduct_panels(world, n):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_xworker, args=(world, port, n, 40, ret), nprocs=world, join=True)
    assert len(ret) == world
    for exch in (False, True):
        edges = sorted((ret[r][exch][0], ret[r][exch][1]) for r in range(world))
        assert edges[0][0] == 0 and edges[-1][1] == n
        for (b0, e0), (b1, e1) in zip(edges, edges[1:]):
            assert e0 == b1
        assert all(ret[r][exch][2] in (True, None) for r in range(world))
        assert any(ret[r][exch][2] for r in range(world))
