"""The two schedules of the transposed-operand int8 kernel (k_gemm_i8_tn; TnSched in csrc/mxa_plan.h: mode 0 = equal pieces in rounds, mode 1 = heads
and tails) on one-copy objects: the integer partial sums are exact, so WHERE a strip's K range is cut must not change one bit of the result.  MXA_TN_SCHED
(read per product: a test / A-B knob) forces a mode where the shape allows it; the shapes here are ones where mode 1 is valid -- several hundred strips of
256 individuals, so that heads, tails and (two digit tiles per pass: 256 resident slots) whole strips all occur.  The host-side invariants of the schedules
(every stage covered once, slots 0 .. pieces - 1 written once) are swept on the CPU: tests/host/plan_sweep.cpp."""
import os

import numpy as np
import pytest

from _util import Oracle

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    import torch
    import miraculix_amd as mx
    mx.load_shared_library()
    return dict(torch=torch, mx=mx, dev=torch.device("cuda", 0))


def _forced(mode, fn):
    old = os.environ.get("MXA_TN_SCHED")
    os.environ["MXA_TN_SCHED"] = str(mode)
    try:
        return fn()
    finally:
        if old is None:
            os.environ.pop("MXA_TN_SCHED", None)
        else:
            os.environ["MXA_TN_SCHED"] = old


@pytest.mark.parametrize("snps,indiv", [(25600, 100000), (20001, 140003), (60000, 70000)])
@pytest.mark.parametrize("n", [1, 2, 4, 6])
def test_both_schedules_give_the_same_bits_and_match_the_oracle(env, snps, indiv, n):
    torch, mx, dev = env["torch"], env["mx"], env["dev"]
    from bench import synth_plink_device
    plink = synth_plink_device(torch, snps, (indiv + 3) // 4, 11 + n, dev)
    f = mx.read_plink.calc_freq(plink, snps, indiv)
    dg = mx.dgemm_compressed
    dg.set_options(use_gpu=True, not_center=False, verbose=0)
    obj = dg.init_compressed(plink, None, snps, indiv, f, n)        # one packed copy: 'N' multiplies from the SNP-major copy
    try:
        g = torch.Generator(device=dev); g.manual_seed(snps + n)
        B = torch.randn((n, snps), dtype=torch.float64, device=dev, generator=g).t()
        C0 = _forced(0, lambda: dg.dgemm_compressed_main(False, obj, B, snps, indiv))
        assert dg.last_path() == "k_gemm_i8"
        C1 = _forced(1, lambda: dg.dgemm_compressed_main(False, obj, B, snps, indiv))
        assert dg.last_path() == "k_gemm_i8"
        Ca = dg.dgemm_compressed_main(False, obj, B, snps, indiv)    # the planner's own choice
        assert torch.equal(C0, C1) and torch.equal(C0, Ca)
        # 48 individuals against the long-double dense oracle (the individuals' SNP rows are extracted from the packed matrix)
        o = Oracle()
        rng = np.random.default_rng(snps)
        ii = np.sort(rng.choice(indiv, 48, replace=False))
        cols = plink[:, torch.from_numpy(ii // 4).to(dev)].cpu().numpy()                        # snps x 48 bytes, one byte per sampled individual
        geno = (cols >> (2 * (ii % 4)).astype(np.uint8)[None, :]) & 3                            # their 2-bit codes, snps x 48
        packed = np.zeros((snps, 12), dtype=np.uint8)
        for q in range(48):
            packed[:, q // 4] |= (geno[:, q] << (2 * (q % 4))).astype(np.uint8)
        prob = dict(snps=snps, indiv=48, plink=packed, plink_t=None, f=f.cpu().numpy())
        Bh = np.ascontiguousarray(B.t().cpu().numpy())                                           # n x snps
        want = o.dgemm_dense(0, prob, Bh, 1)                                                     # n x 48, centred with the full matrix's frequencies
        got = C1[torch.from_numpy(ii).to(dev)].t().cpu().numpy()
        assert np.abs(got - want).max() <= 1e-11 * np.abs(want).max()
    finally:
        dg.free_compressed(obj)
