"""SNP-sharded staging on one GPU: k shard objects built with mxa_plink2compressed_shard from the FULL matrices behave like
the ranks of the multi-GPU job (miraculix_amd/distributed.py): 'N' partials (centring term included) sum to the full
result, 'T' row blocks concatenate to it.  Also the explicit-stream asynchronous device entry."""
import ctypes

import numpy as np
import pytest

from _util import Oracle, make_B, make_problem

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mx():
    import miraculix_amd as m
    m.load_shared_library()
    return m


@pytest.mark.parametrize("world", [2, 3, 8])
@pytest.mark.parametrize("centered", [0, 1])
def test_shards_reproduce_full_result(mx, world, centered):
    from miraculix_amd.distributed import shard_bounds
    o = Oracle()
    snps, indiv, n = 2051, 777, 6
    prob = make_problem(snps, indiv, n, seed=8)
    dg = mx.dgemm_compressed
    dg.set_options(use_gpu=True, not_center=not centered, verbose=0)
    BN, BT = make_B(snps, n, seed=1), make_B(indiv, n, seed=2)
    refN = o.dgemm_dense(0, prob, BN, centered)
    refT = o.dgemm_dense(1, prob, BT, centered)
    CN = np.zeros((indiv, n), order="F")
    CT = np.zeros((snps, n), order="F")
    for r in range(world):
        b, e = shard_bounds(snps, world, r)
        if b == e:
            continue
        obj = dg.init_compressed_shard(prob["plink"], prob["plink_t"], snps, indiv, b, e, prob["f"], n)
        CN += dg.dgemm_compressed_main(False, obj, np.asfortranarray(BN[:, b:e].T), e - b, indiv)      # partial sums incl. centring
        CT[b:e] = dg.dgemm_compressed_main(True, obj, np.asfortranarray(BT.T), e - b, indiv)
        dg.free_compressed(obj)
    assert np.abs(CN.T - refN).max() <= 1e-11 * np.abs(refN).max()
    assert np.abs(CT.T - refT).max() <= 1e-11 * np.abs(refT).max()


def test_explicit_stream_async_entry(mx):
    import torch
    o = Oracle()
    dev = torch.device("cuda", 0)
    prob = make_problem(1500, 640, 12, seed=3)
    dg = mx.dgemm_compressed
    L = mx.check_library_handle()
    dg.set_options(use_gpu=True, not_center=False, verbose=0)
    obj = dg.init_compressed(prob["plink"], prob["plink_t"], 1500, 640, prob["f"], 12)
    s = torch.cuda.Stream(device=dev)
    B = make_B(1500, 12, seed=6)
    with torch.cuda.stream(s):
        Bd = torch.from_numpy(B).to(dev, non_blocking=True).t() * 1.0          # produced on stream s
        Bd = Bd.t().contiguous().t()
        Cd = torch.zeros((12, 640), dtype=torch.float64, device=dev).t()
        rc = L.mxa_dgemm_compressed_device(b"N", obj, 12, ctypes.c_void_p(Bd.data_ptr()), 1500, ctypes.c_void_p(Cd.data_ptr()), 640, ctypes.c_void_p(s.cuda_stream), 0)
        assert rc == 0
        Cd2 = Cd * 2.0                                                           # consumer on the same stream, no host sync in between
    s.synchronize()
    ref = o.dgemm_dense(0, prob, B, 1)
    assert np.abs(Cd2.t().cpu().numpy() / 2.0 - ref).max() <= 1e-11 * np.abs(ref).max()
    dg.free_compressed(obj)
