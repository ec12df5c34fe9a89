"""The two exact engines of the crossproduct (mxa_crossprod.hip): FP4 MFMA (default while sum z z' < 2^24 is guaranteed) and int8 MFMA
(longer inner dimensions, or forced with MXA_XPROD_ENGINE=i8).  Both must reproduce the exact integer oracle bit for bit, including
near the FP4 engine's exactness limit where the fp32 accumulator carries 24 significant bits, and the engine switch must follow
the documented rule (values of 3 present: K < 1 864 135; absent: K < 4 194 304)."""
import os

import numpy as np
import pytest

from _util import Oracle

pytestmark = pytest.mark.gpu


def _xprod(X, k, rows, plink, env=None):
    import miraculix_amd as mx
    mx.load_shared_library()
    old = os.environ.get("MXA_XPROD_ENGINE")
    try:
        if env:
            os.environ["MXA_XPROD_ENGINE"] = env
        else:
            os.environ.pop("MXA_XPROD_ENGINE", None)
        return mx.crossproduct.snp_crossprod(X, k, rows, is_snpmajor=False, is_plink_format=plink)
    finally:
        if old is None:
            os.environ.pop("MXA_XPROD_ENGINE", None)
        else:
            os.environ["MXA_XPROD_ENGINE"] = old


@pytest.mark.parametrize("k,rows,plink", [(1000, 300, True), (4099, 515, True), (777, 260, False), (130, 1, False), (20000, 700, True)])
def test_both_engines_bit_exact(k, rows, plink):
    o = Oracle()
    rng = np.random.default_rng(k)
    X = rng.integers(0, 256, size=(rows, (k + 3) // 4), dtype=np.uint8)      # raw bytes: missing codes / value 3 included
    if k % 4:
        X[:, -1] &= (1 << (2 * (k % 4))) - 1
    ref = o.crossprod_i32(X, k, plink).astype(np.float64)
    assert np.array_equal(_xprod(X, k, rows, plink), ref)
    assert np.array_equal(_xprod(X, k, rows, plink, env="i8"), ref)


def test_fp4_engine_exact_next_to_its_limit():
    """K just below 2^24 / 9 with raw values 0..3 (dense in 3s): sums reach ~16.5 M quarter-units with arbitrary low bits; then a longer K
    takes the int8 engine (same results as the oracle either way)"""
    import torch
    import miraculix_amd as mx
    L = mx.load_shared_library()
    o = Oracle()
    rows = 256
    for k, engine_ms_ratio in ((1_864_000, None), (1_900_000, None)):
        rng = np.random.default_rng(1)
        X = rng.integers(0, 256, size=(rows, k // 4), dtype=np.uint8)
        X[:64] |= 0xC3                                                     # many 3s: rows with large sums
        X[64:66] = 0xFF                                                    # all 3s: 9 K on the diagonal, next to 2^24 for the first K
        ref = o.crossprod_i32(X, k, False).astype(np.float64)
        assert ref.max() == 9.0 * k
        got = _xprod(X, k, rows, False)
        assert np.array_equal(got, ref), k
    # device-resident operands, PLINK data without missings (no 3 after the table): FP4 up to K < 4 194 304
    dev = torch.device("cuda", 0)
    k = 4_000_000
    g = torch.Generator(device=dev); g.manual_seed(3)
    b = torch.randint(0, 256, (rows, k // 4), dtype=torch.uint8, device=dev, generator=g)
    miss = (b & 0x55) & ~((b >> 1) & 0x55)
    Xd = b ^ miss
    Xd[:2] = 0xFF                                                          # all code 11 = value 2: 4 K = 16 000 000 on the diagonal
    M = mx.crossproduct.snp_crossprod(Xd, k, rows, is_snpmajor=False, is_plink_format=True)
    assert float(M[0, 0]) == 4.0 * k and float(M[0, 1]) == 4.0 * k
    sub = Xd[:40].cpu().numpy()
    ref = o.crossprod_i32(sub, k, True).astype(np.float64)
    assert np.array_equal(M[:40, :40].cpu().numpy(), ref)


@pytest.mark.parametrize("plink,k_below,k_above", [(True, 4_194_300, 4_194_304), (False, 1_864_132, 1_864_136)])
def test_fp4_threshold_boundaries_bit_exact(plink, k_below, k_above):
    """The exactness claim of the FP4 engine rests on the fp32 accumulator of v_mfma_scale_f32_32x32x64_f8f6f4 carrying every quarter unit
    up to 2^24 of them.  Probed at the boundary itself: rows that are the largest value throughout (PLINK 11 = 2 without a 3 in the
    matrix: limit 4 K < 2^24; raw 3s: limit 9 K < 2^24) except for a few 1s at the very END of K, so that single quarter units (1 x 1 / 4)
    are added when the running sum already sits at the top of the 24-bit range; plus random rows.  Just below each limit the default
    engine (FP4) must agree with the int8 engine and the closed form bit for bit; just above it the default must still be exact (it has
    switched to int8)."""
    rows = 256
    big = 0xFF                                         # PLINK: four codes 11 (value 2); raw: four 3s
    one = 0xAA if plink else 0x55                      # PLINK 10 -> 1; raw 01 -> 1
    top, lo = (2, 1) if plink else (3, 1)
    for k, below in ((k_below, True), (k_above, False)):
        assert (top * top * k < 2 ** 24) == below
        rng = np.random.default_rng(k)
        X = rng.integers(0, 256, size=(rows, k // 4), dtype=np.uint8)
        if plink:                                      # no missing code 01 anywhere: the staged matrix holds no 3
            miss = (X & 0x55) & ~((X >> 1) & 0x55)
            X ^= miss
        X[:4] = big
        X[1, -2:] = one                                # row 1: the last 8 values are 1
        X[2, -1] = one                                 # row 2: the last 4
        X[3, -3:] = one
        got = _xprod(X, k, rows, plink)
        ref8 = _xprod(X, k, rows, plink, env="i8")
        assert np.array_equal(got, ref8), (k, "default engine vs int8 engine")
        # closed forms of the top-left 4 x 4 block: (k - c) values `top`, c values `lo` at the end
        ones = [0, 8, 4, 12]
        for a in range(4):
            for b in range(4):
                both = min(ones[a], ones[b]); either = max(ones[a], ones[b])
                want = top * top * (k - either) + top * lo * (either - both) + lo * lo * both
                assert got[a, b] == float(want), (k, a, b)


@pytest.mark.parametrize("env", [{"MXA_XPROD_GANG": "0"}, {"MXA_XPROD_GANG": "2"}, {"MXA_XPROD_GANG": "2", "MXA_XPROD_GANG_XCC_MASK": "0"}, {"MXA_XPROD_GANG": "2", "MXA_XPROD_GANG_XCC_MASK": "1"},
                                 {"MXA_XPROD_GANG": "2", "MXA_XPROD_GANG_US": "0"}, {"MXA_XPROD_GANG": "2", "MXA_XPROD_GANG_MID": "1"},
                                 {"MXA_XPROD_GANG": "2", "MXA_XPROD_GANG_MID": "4", "MXA_XPROD_GANG_XCC_MASK": "1"}])
def test_gang_synchronised_kernel_does_not_depend_on_the_xcd_population(env):
    """k_crossprod_gang (one resident workgroup per CU, the workgroups of an XCD advance through that XCD's tile list in step): the result must not
    depend on which XCDs the hardware populated -- with the XCD id masked to one or two values the other lists are stolen --, on the join timeout,
    or on the kernel form at all (MXA_XPROD_GANG=0: one workgroup per tile; 2: the gang form also for launches too short for it to pay, like this one;
    MXA_XPROD_GANG_MID: parts a tile's K range is cut into between meetings of a gang -- default 2 = one meeting half way, 1 = none, 4 = three).  The knobs are read once per process: a child process per setting;
    2304 rows = 45 upper tiles... too few for the per-XCD lists, so 6000 rows (300 tiles, lists of 64 slots) and both engines."""
    import os
    import subprocess
    import sys
    code = """
import sys, numpy as np
sys.path.insert(0, %r); sys.path.insert(0, %r)
import miraculix_amd as mx
from _util import pack_plink
mx.load_shared_library()
rng = np.random.default_rng(3)
rows, k = 6000, 900
Z = rng.integers(0, 3, size=(rows, k)).astype(np.int8)
X = np.ascontiguousarray(pack_plink(Z))
ref = Z.astype(np.float64) @ Z.astype(np.float64).T          # exact: every entry is an integer below 2^53 (an int64 matmul takes numpy five times as long)
import os
for eng in ("f4", "i8"):
    os.environ["MXA_XPROD_ENGINE"] = eng
    M = mx.crossproduct.snp_crossprod(X, k, rows, is_snpmajor=False, is_plink_format=True)
    assert np.array_equal(M, ref), eng
    # the mapped epilogues (GRM / LD) in the same kernel form: fused == the three separate passes, bit for bit
    f = rng.uniform(0.1, 0.5, size=k)
    os.environ["MXA_XPROD_FUSED_POST"] = "1"; G1 = mx.crossproduct.grm(X, k, rows, is_plink_format=True, do_scale=True, allele_freq=f)
    os.environ["MXA_XPROD_FUSED_POST"] = "0"; G0 = mx.crossproduct.grm(X, k, rows, is_plink_format=True, do_scale=True, allele_freq=f)
    assert np.array_equal(G1, G0), eng
    del os.environ["MXA_XPROD_FUSED_POST"]
print("gang ok")
""" % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=dict(os.environ, **env))
    assert r.returncode == 0 and "gang ok" in r.stdout, r.stdout + r.stderr
