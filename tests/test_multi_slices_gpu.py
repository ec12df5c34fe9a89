"""mxa_dgemm_compressed_multi: products on a multi-device object with the operands handed over PER SHARD (every shard reads its own slice
of B and writes its own block of C on its own device; only the indiv x n partial sums cross devices), synchronous and asynchronous, plus
the per-shard report (mxa_multi_get_info / mxa_multi_shard_info) and the switch of the reduction.  On the one-GPU test box the shards
are "virtual" (all on cuda:0): the same host code, worker threads, per-shard streams, copy streams, events and fixed-order reduction run
as on a multi-GPU node.  The last test needs two physical devices and runs on the first box that has them (peer copies, cross-device
event waits, RCCL with more than one rank, the RCCL-vs-peer-to-peer cross-check)."""
import ctypes
import os

import numpy as np
import pytest

from _util import Oracle, make_B, make_problem
from test_multi_gpu import _env, _make

pytestmark = pytest.mark.gpu
RTOL = 1e-11


@pytest.fixture(scope="module")
def mx():
    import miraculix_amd as m
    m.load_shared_library()
    return m


def _cm(torch, a, device):
    """numpy (rows x n) -> column-major torch tensor on `device`"""
    return torch.from_numpy(np.ascontiguousarray(a.T)).to(device).t()


@pytest.mark.parametrize("shards", [2, 3, 8])
@pytest.mark.parametrize("centered", [0, 1])
def test_per_shard_operands_sync_and_async(mx, shards, centered):
    import torch
    o = Oracle()
    snps, indiv, n = 2051, 777, 6
    prob = make_problem(snps, indiv, n, seed=8)
    dg = mx.dgemm_compressed
    dg.set_options(use_gpu=True, not_center=not centered, verbose=0)
    obj = _make(mx, prob, n, shards)
    try:
        bounds = dg.shard_bounds(obj, snps)
        assert len(bounds) == shards and bounds[0][0] == 0 and bounds[-1][1] == snps
        dev = torch.device("cuda", 0)
        BN, BT = make_B(snps, n, seed=1), make_B(indiv, n, seed=2)          # n x k (row j = column j)
        refN = o.dgemm_dense(0, prob, BN, centered)[:, :indiv]
        refT = o.dgemm_dense(1, prob, BT, centered)[:, :snps]
        # the plain entry on the same object: same kernels, same fixed-order reduction -> the per-shard entry must agree bit for bit
        plainN = dg.dgemm_compressed_main(False, obj, np.asfortranarray(BN.T), snps, indiv)
        plainT = dg.dgemm_compressed_main(True, obj, np.asfortranarray(BT.T), snps, indiv)
        ld = max(e - b for b, e in bounds)
        # one leading dimension for all slices of an operand: pad every slice buffer to the largest block
        BN_s = []
        for b, e in bounds:
            buf = torch.full((n, ld), 1e300, dtype=torch.float64, device=dev)
            buf[:, : e - b] = torch.from_numpy(BN[:, b:e]).to(dev)
            BN_s.append(buf.t()[: e - b])
        BT_d = _cm(torch, BT.T, dev)
        for sync in (True, False):
            CN = torch.full((n, indiv), -7.0, dtype=torch.float64, device=dev).t()
            CT_s = [torch.full((n, ld), -7.0, dtype=torch.float64, device=dev).t()[: e - b] for b, e in bounds]
            for rep in range(3 if not sync else 1):     # asynchronous: three steps back to back exercise the buffer-reuse ordering
                dg.dgemm_compressed_multi(False, obj, BN_s, [CN] + [None] * (shards - 1), sync=sync)
                dg.dgemm_compressed_multi(True, obj, [BT_d] + [None if g % 2 else BT_d for g in range(1, shards)], CT_s, sync=sync)
            dg.multi_synchronize(obj)
            gotN = CN.cpu().numpy()
            gotT = np.concatenate([c.cpu().numpy() for c in CT_s])
            assert np.array_equal(gotN, plainN) and np.array_equal(gotT, plainT)
            assert np.abs(gotN.T - refN).max() <= RTOL * np.abs(refN).max()
            assert np.abs(gotT.T - refT).max() <= RTOL * np.abs(refT).max()
            for c, (b, e) in zip(CT_s, bounds):          # rows beyond a block are not touched
                base = c.t()                             # n x ld view of the buffer
                assert bool((base[:, e - b:] == -7.0).all())
        info = dg.multi_info(obj)
        assert info["shards"] == shards and info["devices"] == 1 and info["reduction"] == "p2p-fixed-order"
        assert info["reductions"] == 1 + 1 + 3                                  # plain 'N', sync 'N', three async 'N'
        for g, si in enumerate(info["per_shard"]):
            assert (si["snp_begin"], si["snp_end"]) == bounds[g] and si["peer_to_root"] == -1 and si["pushes"] == 0
            assert si["kernel_launches"] == 2 * (1 + 1 + 3) and si["kernel_ms"] > 0
            # the plain entry staged host B / C through every shard; the per-shard calls with local slices copied nothing, except the
            # shards that were told to read B of the 'T' product from shard 0 (same device here: still local)
            assert si["in_copies"] == 2 and si["out_copies"] == 1
        assert dg.multi_info(obj, reset=True)["reductions"] == 5
        assert dg.multi_info(obj)["reductions"] == 0
        assert dg.multi_set_reduction(obj, "rccl") is False                     # several shards share the device: not applicable, unchanged
        assert dg.multi_info(obj)["reduction"] == "p2p-fixed-order"
    finally:
        dg.free_compressed(obj)


def test_per_shard_entry_argument_errors(mx):
    import torch
    prob = make_problem(400, 90, 2, seed=3)
    dg = mx.dgemm_compressed
    L = mx.check_library_handle()
    dg.set_options(use_gpu=True, not_center=True, verbose=0)
    obj = _make(mx, prob, 2, 2)
    single = dg.init_compressed(prob["plink"], prob["plink_t"], 400, 90, prob["f"], 2)
    try:
        dev = torch.device("cuda", 0)
        B = torch.zeros((2, 200), dtype=torch.float64, device=dev).t()
        C = torch.zeros((2, 90), dtype=torch.float64, device=dev).t()
        with pytest.raises(RuntimeError):
            dg.dgemm_compressed_multi(False, obj, [B, None], [C, None])          # shard 1 has no B slice
        with pytest.raises(RuntimeError):
            dg.dgemm_compressed_multi(True, obj, [C, C], [B, None])              # shard 1 has no C block
        Bp = (ctypes.c_void_p * 1)(B.data_ptr()); Cp = (ctypes.c_void_p * 1)(C.data_ptr())
        assert L.mxa_dgemm_compressed_multi(b"N", single, 2, Bp, 200, Cp, 90, 1) == 1   # not a multi-device object
        assert L.mxa_multi_synchronize(single) == 1 and L.mxa_multi_set_reduction(single, 0) == 1
        dg.dgemm_compressed_multi(False, obj, [B, B], [C, None])                 # and the object still works
    finally:
        dg.free_compressed(obj)
        dg.free_compressed(single)


def test_rccl_reduction_is_cross_checked_on_first_use(mx):
    """one-shard multi object (MXA_FORCE_MULTI): the RCCL reduction can be switched on at run time; its first product also runs the
    peer-to-peer fixed-order reduction and compares (one rank: identical)"""
    o = Oracle()
    snps, indiv, n = 1200, 333, 3
    prob = make_problem(snps, indiv, n, seed=6)
    dg = mx.dgemm_compressed
    dg.set_options(use_gpu=True, not_center=False, verbose=0)
    obj = _make(mx, prob, n, 1, MXA_FORCE_MULTI=1)
    try:
        B = make_B(snps, n, seed=1)
        ref = o.dgemm_dense(0, prob, B, 1)[:, :indiv]
        C0 = dg.dgemm_compressed_main(False, obj, np.asfortranarray(B.T), snps, indiv)
        assert dg.multi_info(obj)["rccl_checked"] == 0
        assert dg.multi_set_reduction(obj, "rccl") is True
        C1 = dg.dgemm_compressed_main(False, obj, np.asfortranarray(B.T), snps, indiv)
        info = dg.multi_info(obj)
        assert info["reduction"] == "rccl" and info["rccl_checked"] == 1 and info["rccl_vs_p2p_max_rel_diff"] == 0.0
        assert np.array_equal(C0, C1) and np.abs(C1.T - ref).max() <= RTOL * np.abs(ref).max()
        assert dg.multi_set_reduction(obj, "p2p") is True
        assert np.array_equal(dg.dgemm_compressed_main(False, obj, np.asfortranarray(B.T), snps, indiv), C0)
    finally:
        dg.free_compressed(obj)


def test_operand_produced_on_the_default_stream_just_before_the_call(mx):
    """B is the output of work that is still running on the device's default stream when the product is issued (a long chain of
    PyTorch kernels): the object's blocking streams order themselves behind it.  (Across devices the library waits for the producing
    device's default stream explicitly -- exercised by the two-device test.)"""
    import torch
    dev = torch.device("cuda", 0)
    prob = make_problem(4096, 1024, 8, seed=11)
    dg = mx.dgemm_compressed
    dg.set_options(use_gpu=True, not_center=True, verbose=0)
    obj = _make(mx, prob, 8, 3)
    try:
        g = torch.Generator(device=dev); g.manual_seed(1)
        seed_mat = torch.randn((2048, 2048), dtype=torch.float64, device=dev, generator=g) / 45.0
        base = torch.randn((8, 4096), dtype=torch.float64, device=dev, generator=g)
        torch.cuda.synchronize()
        x = seed_mat
        for _ in range(40):                      # ~40 fp64 2048^3 products: tens of milliseconds of queued work
            x = x @ seed_mat
        B = (base + x[0, 0] * 0.0).t()           # 4096 x 8, column-major; depends on the end of the chain
        C = dg.dgemm_compressed_main(False, obj, B, 4096, 1024)
        torch.cuda.synchronize()
        C2 = dg.dgemm_compressed_main(False, obj, B, 4096, 1024)
        assert torch.equal(C, C2) and bool(torch.isfinite(C).all())
    finally:
        dg.free_compressed(obj)


def test_two_physical_devices(mx):
    """Runs only where two GPUs are visible: shards on distinct devices (peer pushes over xGMI, cross-device event waits), operands
    per device and on one device (hub), an operand still being produced on another device's default stream, the RCCL reduction with two
    ranks and its cross-check against the peer-to-peer one."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two visible GPUs")
    o = Oracle()
    snps, indiv, n = 40_004, 3_001, 12
    prob = make_problem(snps, indiv, n, seed=2)
    dg = mx.dgemm_compressed
    dg.set_options(use_gpu=True, not_center=False, verbose=0)
    obj = _make(mx, prob, n, 2)
    try:
        info = dg.multi_info(obj)
        assert info["devices"] == 2 and [s["device"] for s in info["per_shard"]] == [0, 1]
        bounds = dg.shard_bounds(obj, snps)
        BN, BT = make_B(snps, n, seed=1), make_B(indiv, n, seed=2)
        refN = o.dgemm_dense(0, prob, BN, 1)[:, :indiv]
        refT = o.dgemm_dense(1, prob, BT, 1)[:, :snps]
        plainN = dg.dgemm_compressed_main(False, obj, np.asfortranarray(BN.T), snps, indiv)
        assert np.abs(plainN.T - refN).max() <= RTOL * np.abs(refN).max()
        ld = max(e - b for b, e in bounds)
        devs = [torch.device("cuda", g) for g in range(2)]
        BN_s, CT_s = [], []
        for (b, e), d in zip(bounds, devs):
            buf = torch.zeros((n, ld), dtype=torch.float64, device=d)
            buf[:, : e - b] = torch.from_numpy(BN[:, b:e]).to(d)
            BN_s.append(buf.t()[: e - b])
            CT_s.append(torch.zeros((n, ld), dtype=torch.float64, device=d).t()[: e - b])
        BT_s = [_cm(torch, BT.T, d) for d in devs]
        CN = torch.zeros((n, indiv), dtype=torch.float64, device=devs[0]).t()
        for _ in range(3):
            dg.dgemm_compressed_multi(False, obj, BN_s, [CN, None], sync=False)
            dg.dgemm_compressed_multi(True, obj, BT_s, CT_s, sync=False)
        dg.multi_synchronize(obj)
        assert np.array_equal(CN.cpu().numpy(), plainN)
        gotT = np.concatenate([c.cpu().numpy() for c in CT_s])
        assert np.abs(gotT.T - refT).max() <= RTOL * np.abs(refT).max()
        # hub: B of the 'N' product lives on device 1 and is still being produced there when the call is issued
        with torch.cuda.device(1):
            seed_mat = torch.randn((2048, 2048), dtype=torch.float64, device=devs[1]) / 45.0
            x = seed_mat
            for _ in range(40):
                x = x @ seed_mat
            Bhub = (torch.from_numpy(BN).to(devs[1]) + x[0, 0] * 0).t()
        hub = dg.dgemm_compressed_main(False, obj, Bhub, snps, indiv)
        assert np.array_equal(hub.cpu().numpy(), plainN)
        # a hub operand large enough for the K-range pipeline (> 32 MB): it streams from device 1 over xGMI behind the products of both shards
        n_big = 128
        BNb = make_B(snps, n_big, seed=5)
        with torch.cuda.device(1):
            Bbig = torch.from_numpy(BNb).to(devs[1]).t()
        big_hub = dg.dgemm_compressed_main(False, obj, Bbig, snps, indiv)
        big_host = dg.dgemm_compressed_main(False, obj, np.asfortranarray(BNb.T), snps, indiv)
        assert np.array_equal(big_hub.cpu().numpy(), big_host)
        # RCCL with two ranks, cross-checked against the peer-to-peer reduction on its first product
        assert dg.multi_set_reduction(obj, "rccl") is True
        rc = dg.dgemm_compressed_main(False, obj, np.asfortranarray(BN.T), snps, indiv)
        info = dg.multi_info(obj)
        assert info["rccl_checked"] == 1 and 0.0 <= info["rccl_vs_p2p_max_rel_diff"] <= 1e-13
        assert np.abs(rc.T - refN).max() <= RTOL * np.abs(refN).max()
        assert all(s["peer_to_root"] in (-1, 0, 1) for s in info["per_shard"])
    finally:
        dg.free_compressed(obj)


@pytest.mark.parametrize("shards", [2, 8])
def test_async_chain_T_reads_the_result_of_the_preceding_N(mx, shards):
    """products issued back to back with sync = 0 are ordered like calls on one stream (include/miraculix_amd.h): an 'N' product into C followed at
    once by a 'T' product whose B IS that C (the step Z^T (Z x)) -- the reduction delivers C on the root stream, the shard streams must wait for
    it (advisor finding of round 3: the 'T' branch did not).  A big enough problem that the reduction is still running when 'T' is enqueued;
    the asynchronous chain must equal the synchronous one bit for bit, five times in a row."""
    import torch
    from bench import synth_genotypes_device
    snps, indiv, n = 60_000, 20_000, 8
    dev = torch.device("cuda", 0)
    plink = synth_genotypes_device(torch, snps, indiv, 18, dev)                 # generated, transposed and counted on the device: seconds instead of a minute of numpy
    plink_t = mx.compressed_operations.transpose_genotype_matrix(plink, snps, indiv)
    prob = dict(snps=snps, indiv=indiv, plink=plink, plink_t=plink_t, f=mx.read_plink.calc_freq(plink, snps, indiv))
    dg = mx.dgemm_compressed
    dg.set_options(use_gpu=True, not_center=False, verbose=0)
    obj = _make(mx, prob, n, shards)
    try:
        bounds = dg.shard_bounds(obj, snps)
        ld = max(e - b for b, e in bounds)
        g = torch.Generator(device=dev); g.manual_seed(2)
        X = torch.randn((n, snps), dtype=torch.float64, device=dev, generator=g)
        X_s = []
        for b, e in bounds:
            buf = torch.zeros((n, ld), dtype=torch.float64, device=dev)
            buf[:, : e - b] = X[:, b:e]
            X_s.append(buf.t()[: e - b])

        def chain(sync):
            CN = torch.full((n, indiv), float("nan"), dtype=torch.float64, device=dev).t()
            CT_s = [torch.full((n, ld), float("nan"), dtype=torch.float64, device=dev).t()[: e - b] for b, e in bounds]
            dg.dgemm_compressed_multi(False, obj, X_s, [CN] + [None] * (shards - 1), sync=sync)
            dg.dgemm_compressed_multi(True, obj, [CN] * shards, CT_s, sync=sync)
            dg.multi_synchronize(obj)
            return CN, torch.cat([c for c in CT_s])

        CN0, CT0 = chain(True)
        assert bool(torch.isfinite(CT0).all())
        for _ in range(5):
            CN1, CT1 = chain(False)
            assert torch.equal(CN1, CN0) and torch.equal(CT1, CT0)
    finally:
        dg.free_compressed(obj)
