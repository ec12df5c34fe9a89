"""GPU: sparse_times_plink through the C ABI against the oracle and the reference's golden vectors."""
import os

import numpy as np
import pytest

from _util import Oracle, make_B, make_problem, random_csr

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "sparse_golden.npz")
RTOL = 1e-13   # sums of at most a few dozen products of O(1) numbers


@pytest.fixture(scope="module")
def dg():
    import miraculix_amd as m
    m.load_shared_library()
    return m.dgemm_compressed


def test_sparse_golden(dg):
    g = np.load(GOLD)
    for name in g["names"]:
        snps, indiv, ldc, nidx = (int(x) for x in g[f"{name}/dims"])
        plink, plink_t = np.ascontiguousarray(g[f"{name}/plink"]), np.ascontiguousarray(g[f"{name}/plink_t"])
        for tc in ("N", "T"):
            entries = snps if tc == "T" else indiv
            out = np.full((ldc, entries), -777.0, order="F")
            C = dg.sparse_times_plink(tc == "T", plink, plink_t, snps, indiv, g[f"{name}/ia{tc}"], g[f"{name}/ja{tc}"], g[f"{name}/a{tc}"], ldc=ldc, out=out)
            ref = g[f"{name}/C{tc}"].T   # (ldc, entries)
            assert np.abs(C - ref).max() <= RTOL * max(1.0, np.abs(ref).max()), (name, tc)
            assert np.all(C[nidx:, :] == 0.0)


@pytest.mark.parametrize("snps,indiv,nidx,max_nnz", [(3001, 2050, 100, 25), (513, 5000, 1, 513), (5000, 517, 300, 3), (2048, 1024, 17, 200)])
def test_sparse_vs_oracle(dg, snps, indiv, nidx, max_nnz):
    o = Oracle()
    prob = make_problem(snps, indiv, 1, seed=snps + indiv, missing_frac=0.05)
    for tc in (False, True):
        rows, entries = (indiv, snps) if tc else (snps, indiv)
        P = prob["plink_t"] if tc else prob["plink"]
        ia, ja, a = random_csr(nidx, rows, max_nnz, seed=7 + tc)
        ref = o.sparse_times_plink(P, rows, entries, ia, ja, a).T
        C = dg.sparse_times_plink(tc, prob["plink"] if not tc else None, prob["plink_t"] if tc else None, snps, indiv, ia, ja, a)
        assert C.shape == (nidx, entries)
        assert np.abs(C - ref).max() <= RTOL * max(1.0, np.abs(ref).max())


def test_sparse_device_operands_and_errors(dg):
    import torch
    import miraculix_amd as m
    o = Oracle()
    prob = make_problem(1200, 640, 1, seed=3)
    ia, ja, a = random_csr(40, 1200, 20, seed=1)
    ref = o.sparse_times_plink(prob["plink"], 1200, 640, ia, ja, a).T
    d_plink = torch.from_numpy(prob["plink"]).cuda()
    d_C = torch.full((640, 40), -1.0, dtype=torch.float64, device="cuda").t()   # column-major 40 x 640
    C = dg.sparse_times_plink(False, d_plink, None, 1200, 640, ia, ja, a, out=d_C)
    assert np.abs(C.cpu().numpy() - ref).max() <= RTOL * np.abs(ref).max()
    # empty sparse matrix rows give zero rows; an out-of-range column index is an error and leaves C unwritten
    ia0 = np.zeros(6, np.int32)
    C0 = dg.sparse_times_plink(False, prob["plink"], None, 1200, 640, ia0, np.zeros(0, np.int32), np.zeros(0))
    assert C0.shape == (5, 640) and np.all(C0 == 0.0)
    bad = ja.copy(); bad[3] = 1200
    with pytest.raises(RuntimeError):
        dg.sparse_times_plink(False, prob["plink"], None, 1200, 640, ia, bad, a)
    L = m.lib.check_library_handle()
    assert L.mxa_last_error() != 0
    dg.sparse_times_plink(False, prob["plink"], None, 1200, 640, ia, ja, a)      # the status describes the most recent call
    assert L.mxa_last_error() == 0


@pytest.mark.parametrize("snps,indiv,n", [(1003, 501, 5), (2048, 640, 32), (777, 1301, 1)])
@pytest.mark.parametrize("trans", [0, 1])
@pytest.mark.parametrize("centered", [0, 1])
def test_dgemm_plink_documented_semantics(snps, indiv, n, trans, centered):
    """dgemm_plink (reference: src/miraculix/5codesAPI.c:112-130): one product straight from the PLINK matrices.  PARITY UNPINNED -- the reference aborts
    unconditionally in this entry (f != NULL: 5codesChar.cc:511-513 BUG; f == NULL: plink256.cc:332 BUG), so the check is the documented semantics against the
    dense oracle: 'N' = Zc B from plink_transposed alone, 'T' = Zc^T B from plink alone, f NULL = uncentred / given = centred, whatever the options say; no
    object is left behind and the options singleton is restored."""
    import miraculix_amd as mx
    from _util import elementwise_bound
    mx.load_shared_library()
    dg = mx.dgemm_compressed
    o = Oracle()
    prob = make_problem(snps, indiv, n, seed=31 + n, missing_frac=0.03)
    k, m = (indiv, snps) if trans else (snps, indiv)
    B = make_B(k, n, seed=8)
    dg.set_options(use_gpu=True, not_center=bool(centered), verbose=0)          # the OPPOSITE of what the call asks for: dgemm_plink decides by f alone
    C = dg.dgemm_plink(bool(trans), prob["plink"] if trans else None, None if trans else prob["plink_t"], snps, indiv, prob["f"] if centered else None, np.asfortranarray(B.T))
    ref = o.dgemm_dense(trans, prob, B, centered)[:, :m]
    assert C.shape == (m, n)
    assert np.abs(C.T - ref).max() <= 1e-11 * np.abs(ref).max()
    assert np.all(np.abs(C.T - ref) <= elementwise_bound(o, trans, prob, B, centered))
    # the options singleton is as the caller left it: a plain object multiplies with not_center = centered
    obj = dg.init_compressed(prob["plink"], prob["plink_t"], snps, indiv, prob["f"], n)
    try:
        C2 = dg.dgemm_compressed_main(bool(trans), obj, np.asfortranarray(B.T), snps, indiv)
        ref2 = o.dgemm_dense(trans, prob, B, 0 if centered else 1)[:, :m]
        assert np.abs(C2.T - ref2).max() <= 1e-11 * np.abs(ref2).max()
    finally:
        dg.free_compressed(obj)
