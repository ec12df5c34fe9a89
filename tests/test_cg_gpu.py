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
