"""Opt-in engine 'i8-guarded' (include/miraculix_amd.h, mxa_set_engine(5)): the default engine's guarded exact int8 route for EVERY n -- the columns go in
balanced chunks of at most six, every chunk with its own verdict formed on the device (exact with the digits of two tiles / of three tiles / not exact -> the
fp64 chains of that chunk), nothing read back by the host.  Element-wise against the long-double oracle on the adversarial inputs of test_small_n_gpu.py (outputs
that see only the small entries of B): |error| <= 3.02 (S - 1) 2^-53 sum |z b| with S <= 24 where the int8 classes apply, the fp64-chain bound K 2^-53 sum |z b|
where a chunk falls back; one-copy (default) and two-copy objects, both products, centred and not, bitwise repeatable, mixed verdicts inside one product."""
import os

import numpy as np
import pytest

from _util import Oracle, make_B, make_problem
from test_small_n_gpu import _adversarial_problem, _wide_B

pytestmark = pytest.mark.gpu
U = 2.0 ** -53


@pytest.fixture(scope="module")
def mx():
    import miraculix_amd as m
    m.load_shared_library()
    prev = m.dgemm_compressed.set_engine("i8-guarded")
    yield m
    m.dgemm_compressed.set_engine(prev)


def _obj(mx, prob, n, copies):
    os.environ["MXA_SINGLE_ORIENTATION"] = "1" if copies == 1 else "0"
    try:
        obj = mx.dgemm_compressed.init_compressed(prob["plink"], prob["plink_t"], prob["snps"], prob["indiv"], prob["f"], n)
    finally:
        os.environ.pop("MXA_SINGLE_ORIENTATION", None)
    assert mx.dgemm_compressed.single_orientation(obj) == (1 if copies == 1 else 0)
    return obj


def _run(mx, obj, prob, trans, B):
    return mx.dgemm_compressed.dgemm_compressed_main(bool(trans), obj, np.asfortranarray(B.T), prob["snps"], prob["indiv"]).T   # n x m


@pytest.mark.parametrize("n,decades", [(7, 3), (8, 12), (10, 5), (13, 16), (17, 4), (32, 6), (40, 10)])
@pytest.mark.parametrize("copies", [1, 2])
def test_every_chunk_exact_and_within_the_int8_bound(mx, n, decades, copies):
    o = Oracle()
    snps, indiv = 3001, 1037
    prob = _adversarial_problem(snps, indiv, seed=11)
    dg = mx.dgemm_compressed
    dg.set_options(use_gpu=True, not_center=True, verbose=0)
    obj = _obj(mx, prob, n, copies)
    try:
        for trans in (0, 1):
            k, m = (indiv, snps) if trans else (snps, indiv)
            B = _wide_B(k, n, decades, seed=3 + trans, big_every=5 if trans else 7)
            C = _run(mx, obj, prob, trans, B)
            assert dg.last_path() == "k_gemm_i8"
            ref = o.dgemm_dense(trans, prob, B, 0)[:, :m]
            abssum = o.dgemm_dense(trans, prob, np.abs(B), 0)[:, :m]
            err = np.abs(C - ref)
            bound = 3.02 * 23 * U * abssum
            assert np.all(err <= bound + 1e-300), (trans, float((err / np.maximum(bound, 1e-300)).max()))
            small = abssum < 1e-3 * abssum.max()
            if small.any():
                assert np.all(err[small] <= 1e-13 * abssum[small])
            assert np.array_equal(C, _run(mx, obj, prob, trans, B))
        # centred
        dg.set_options(use_gpu=True, not_center=False, verbose=0)
        B = _wide_B(snps, n, decades, seed=3, big_every=7)
        Cc = _run(mx, obj, prob, 0, B)
        refc = o.dgemm_dense(0, prob, B, 1)[:, :indiv]
        assert np.abs(Cc - refc).max() <= 1e-11 * np.abs(refc).max()
    finally:
        dg.free_compressed(obj)


@pytest.mark.parametrize("copies", [1, 2])
def test_mixed_verdicts_inside_one_product(mx, copies):
    """n = 16 = chunks of 6 + 5 + 5 columns: the first chunk exact with few digits, the second needs the larger digit count (an entry 60 binades below its
    column's maximum), the third is not representable (150 binades, and an inf in its last column): int8 / int8 / fp64 chains in ONE product"""
    o = Oracle()
    snps, indiv, n = 2600, 900, 16
    prob = make_problem(snps, indiv, n, seed=5)
    dg = mx.dgemm_compressed
    dg.set_options(use_gpu=True, not_center=True, verbose=0)
    obj = _obj(mx, prob, n, copies)
    try:
        for trans in (0, 1):
            k, m = (indiv, snps) if trans else (snps, indiv)
            B = np.random.default_rng(6 + trans).standard_normal((n, k))
            B[7, 3] = 2.0 ** -60
            B[12, 5] = 1e-45
            C = _run(mx, obj, prob, trans, B)
            assert dg.last_path() == "k_small_n_fp64"          # the verdict of the LAST chunk
            ref = o.dgemm_dense(trans, prob, B, 0)[:, :m]
            for j in range(n):
                assert np.abs(C[j] - ref[j]).max() <= 1e-11 * np.abs(ref[j]).max(), (trans, j)
            B[15, 7] = np.inf
            C = _run(mx, obj, prob, trans, B)
            assert not np.isfinite(C[15]).all()
            for j in range(15):
                assert np.abs(C[j] - ref[j]).max() <= 1e-11 * np.abs(ref[j]).max(), (trans, j)
    finally:
        dg.free_compressed(obj)


def test_integer_B_is_bit_exact_and_short_k_takes_the_fp64_path(mx):
    rng = np.random.default_rng(3)
    prob = make_problem(3000, 400, 20, seed=11)
    dg = mx.dgemm_compressed
    dg.set_options(use_gpu=True, not_center=True, verbose=0)
    obj = dg.init_compressed(prob["plink"], prob["plink_t"], 3000, 400, prob["f"], 20)
    try:
        B = rng.integers(-(2 ** 20), 2 ** 20, size=(20, 3000)).astype(np.float64)
        C = _run(mx, obj, prob, 0, B)
        assert dg.last_path() == "k_gemm_i8"
        assert np.array_equal(C.T, (prob["Z"].astype(np.int64) @ B.T.astype(np.int64)).astype(np.float64))
    finally:
        dg.free_compressed(obj)
    prob = make_problem(100, 333, 9, seed=2)          # K = 100 < 128 for 'N'
    obj = dg.init_compressed(prob["plink"], prob["plink_t"], 100, 333, prob["f"], 9)
    try:
        B = make_B(100, 9, seed=2)[:, :100]
        C = _run(mx, obj, prob, 0, B)
        assert dg.last_path() == "k_gemm"
        ref = Oracle().dgemm_dense(0, prob, B, 0)[:, :333]
        assert np.abs(C - ref).max() <= 1e-11 * np.abs(ref).max()
    finally:
        dg.free_compressed(obj)


def test_no_host_wait_with_device_operands(mx):
    """the asynchronous entry must return while work queued BEFORE it is still running (the technique of test_async_gpu.py: a ~1 s spin kernel on the default
    stream, which the object's blocking stream is ordered behind): no product of this engine waits for the host, whatever n"""
    import ctypes
    import time
    import torch
    L = mx.check_library_handle()
    dg = mx.dgemm_compressed
    o = Oracle()
    snps, indiv = 3001, 1200
    prob = make_problem(snps, indiv, 40, seed=5, missing_frac=0.03)
    dg.set_options(use_gpu=True, not_center=False, verbose=0)
    obj = dg.init_compressed(prob["plink"], None, snps, indiv, prob["f"], 40)
    dev = torch.device("cuda", 0)
    ops = {}
    try:
        for n in (7, 12, 23, 40):
            for trans in (0, 1):
                k, m = (indiv, snps) if trans else (snps, indiv)
                B = make_B(k, n, seed=3 * n + trans)
                ops[(n, trans)] = (B, torch.from_numpy(np.ascontiguousarray(B)).to(dev), torch.full((n, m), -7.0, dtype=torch.float64, device=dev), k, m)

        def issue_all():
            for (n, trans), (B, Bd, Cd, k, m) in ops.items():
                assert L.mxa_dgemm_compressed_device(b"T" if trans else b"N", obj, n, ctypes.c_void_p(Bd.data_ptr()), k, ctypes.c_void_p(Cd.data_ptr()), m, None, 0) == 0

        issue_all()                     # warm-up: every workspace has its final size
        torch.cuda.synchronize()
        assert dg.last_path() == "k_gemm_i8"
        for _, _, Cd, _, _ in ops.values():
            Cd.fill_(-7.0)
        torch.cuda.synchronize()
        done = torch.cuda.Event()
        t0 = time.perf_counter()
        torch.cuda._sleep(int(2.0e9))
        done.record()
        issue_all()
        t_issue = time.perf_counter() - t0
        assert not done.query(), f"the asynchronous entries blocked: issuing took {t_issue:.3f} s"
        assert t_issue < 0.5, t_issue
        torch.cuda.synchronize()
        for (n, trans), (B, Bd, Cd, k, m) in ops.items():
            ref = o.dgemm_dense(trans, prob, B, 1)[:, :m]
            assert np.abs(Cd.cpu().numpy() - ref).max() <= 1e-11 * np.abs(ref).max(), (n, trans)
    finally:
        dg.free_compressed(obj)
