"""Single-orientation objects (round 4; MXA_SINGLE_ORIENTATION=1 when plink2compressed runs, or by itself when two packed copies do not fit the device): only the SNP-major copy is stored; 'T' products run in the
plain form, 'N' products in the transposed-operand forms (k_gemm<..., TR>, k_gemm_i8_tn) on the same copy (halves the HBM
footprint and the staging upload).  Every n, both products, centred and not, missing codes, ragged sizes, padded leading dimensions, the range and
exactness fallbacks, mxa_gram_matvec, SNP shards behind MIRACULIX_NUM_GPUS, staging from a .bed file -- against the long-double oracle, against the
two-copy object (bit-identical on the fp64 MFMA path and for integer-valued B), the device memory the object holds, and the automatic case under a
memory squeeze (a ballast tensor fills the device)."""
import ctypes
import os

import numpy as np
import pytest

from _util import Oracle, make_B, make_problem

pytestmark = pytest.mark.gpu
RTOL = 1e-11


@pytest.fixture(scope="module")
def mx():
    import miraculix_amd as m
    m.load_shared_library()
    return m


class _single:
    def __enter__(self):
        os.environ["MXA_SINGLE_ORIENTATION"] = "1"

    def __exit__(self, *a):
        os.environ.pop("MXA_SINGLE_ORIENTATION", None)


def _make(mx, prob, n, single, plink_t="given"):
    dg = mx.dgemm_compressed
    if single:
        with _single():
            obj = dg.init_compressed(prob["plink"], None if plink_t is None else prob["plink_t"], prob["snps"], prob["indiv"], prob["f"], n)
    else:
        os.environ["MXA_SINGLE_ORIENTATION"] = "0"            # both copies: the opt-in since round 5
        try:
            obj = dg.init_compressed(prob["plink"], prob["plink_t"], prob["snps"], prob["indiv"], prob["f"], n)
        finally:
            os.environ.pop("MXA_SINGLE_ORIENTATION", None)
    L = mx.check_library_handle()
    L.mxa_single_orientation.argtypes = [ctypes.c_void_p]
    assert L.mxa_single_orientation(obj) == (1 if single else 0)
    return obj


@pytest.mark.parametrize("snps,indiv", [(2051, 777), (1003, 130), (700, 3001)])
@pytest.mark.parametrize("n", [1, 2, 3, 5, 6, 7, 8, 10, 11, 16, 32, 33, 40])
def test_every_n_both_products(mx, snps, indiv, n):
    o = Oracle()
    prob = make_problem(snps, indiv, n, seed=snps + 7 * n, missing_frac=0.04)
    dg = mx.dgemm_compressed
    one, two = _make(mx, prob, n, True, plink_t=None), _make(mx, prob, n, False)
    try:
        for centered in (0, 1):
            dg.set_options(use_gpu=True, not_center=not centered, verbose=0)
            for trans in (0, 1):
                k, m = (indiv, snps) if trans else (snps, indiv)
                B = make_B(k, n, seed=3 + centered + 2 * trans)
                C1 = dg.dgemm_compressed_main(bool(trans), one, np.asfortranarray(B.T), snps, indiv)
                path1 = dg.last_path()
                C2 = dg.dgemm_compressed_main(bool(trans), two, np.asfortranarray(B.T), snps, indiv)
                path2 = dg.last_path()
                ref = o.dgemm_dense(trans, prob, B, centered)[:, :m]
                assert np.abs(C1.T - ref).max() <= RTOL * np.abs(ref).max()
                assert path1 == path2                                        # the same route as a two-copy object: 'N' with n <= 6 and the peeled columns on the transposed int8 kernel
                if path1 == "k_gemm" and path2 == "k_gemm" and n % 4 == 0:
                    assert np.array_equal(C1, C2)                            # same plan, same sums
                Bi = np.round(B * 64.0)
                if not centered:                                             # integer-valued operands: exact on every route
                    assert np.array_equal(dg.dgemm_compressed_main(bool(trans), one, np.asfortranarray(Bi.T), snps, indiv),
                                          dg.dgemm_compressed_main(bool(trans), two, np.asfortranarray(Bi.T), snps, indiv))
    finally:
        dg.free_compressed(one)
        dg.free_compressed(two)


def test_fallbacks_ld_padding_and_gram(mx):
    """n = 1 with a column the int8 digits cannot hold ('N': the transposed MFMA tile with plain operands behind the device flag); a column beyond the
    denormal-operand range at n = 8; padded Ldb / Ldc with poison; mxa_gram_matvec = 'T' then 'N'"""
    o = Oracle()
    snps, indiv = 1900, 823
    prob = make_problem(snps, indiv, 8, seed=12)
    dg = mx.dgemm_compressed
    L = mx.check_library_handle()
    dg.set_options(use_gpu=True, not_center=False, verbose=0)
    obj = _make(mx, prob, 8, True)
    try:
        for trans in (0, 1):
            k, m = (indiv, snps) if trans else (snps, indiv)
            B = make_B(k, 1, seed=4 + trans)
            B[0, ::3] *= 1e-80                                               # 265 binades: the exactness guard declines
            C = dg.dgemm_compressed_main(bool(trans), obj, np.asfortranarray(B.T), snps, indiv)
            assert dg.last_path() == "k_small_n_fp64"                                 # "the fp64 fallback ran" (mxa_last_path reads the device flag)
            ref = o.dgemm_dense(trans, prob, B, 1)[:, :m]
            err, abssum = np.abs(C.T - ref), o.dgemm_dense(trans, prob, np.abs(B), 0)[:, :m]
            assert np.all(err <= k * 2.0 ** -52 * abssum + 8 * 2.0 ** -53 * np.abs(ref - o.dgemm_dense(trans, prob, B, 0)[:, :m]) + 1e-300)
            ldb, ldc = k + 3, m + 5
            B8 = make_B(k, 8, seed=6 + trans, ldb=ldb)
            B8[2, :k:3] *= 1e-300
            B8[2, 1:k:3] *= 1e+250
            C8 = np.full((8, ldc), -777.0)
            L.dgemm_compressed(b"T" if trans else b"N", obj, 8, B8.ctypes.data_as(ctypes.c_void_p), ldb, C8.ctypes.data_as(ctypes.c_void_p), ldc)
            assert L.mxa_last_range_fallback(obj) == 1 and np.all(C8[:, m:] == 0.0)
            ref8 = o.dgemm_dense(trans, prob, B8, 1, ldc=ldc)
            a8 = o.dgemm_dense(trans, prob, np.abs(B8), 0, ldc=ldc)[:, :m]
            assert np.all(np.abs(C8[:, :m] - ref8[:, :m]) <= k * 2.0 ** -52 * a8 + 8 * 2.0 ** -53 * np.abs(ref8[:, :m]) + 1e-300)
        V = make_B(indiv, 2, seed=9)
        G = dg.gram_matvec(obj, np.asfortranarray(V.T), snps, indiv)
        T = dg.dgemm_compressed_main(True, obj, np.asfortranarray(V.T), snps, indiv)
        N = dg.dgemm_compressed_main(False, obj, T, snps, indiv)
        assert np.array_equal(G, N)
        refT = o.dgemm_dense(1, prob, V, 1)[:, :snps]
        refG = o.dgemm_dense(0, prob, np.ascontiguousarray(refT), 1)[:, :indiv]
        assert np.abs(G.T - refG).max() <= RTOL * np.abs(refG).max()
    finally:
        dg.free_compressed(obj)


def test_half_the_device_memory_shards_and_bed(mx, tmp_path):
    """the object holds one packed copy (device memory measured around its creation); SNP shards behind MIRACULIX_NUM_GPUS and staging from a .bed file
    honour the mode too"""
    import torch
    o = Oracle()
    snps, indiv, n = 16_000, 12_000, 4
    prob = make_problem(snps, indiv, n, seed=2)
    dg = mx.dgemm_compressed
    dg.set_options(use_gpu=True, not_center=False, verbose=0)
    torch.cuda.synchronize()
    used = []
    for single in (False, True):
        free0 = torch.cuda.mem_get_info()[0]
        obj = _make(mx, prob, n, single, plink_t=None)
        used.append(free0 - torch.cuda.mem_get_info()[0])
        dg.free_compressed(obj)
    packed = snps * indiv / 4
    assert used[0] - used[1] >= 0.9 * packed and used[1] <= used[0] - 0.9 * packed    # one orientation less
    B = make_B(snps, n, seed=1)
    ref = o.dgemm_dense(0, prob, B, 1)[:, :indiv]
    os.environ["MIRACULIX_NUM_GPUS"] = "3"
    try:
        with _single():
            sh = dg.init_compressed(prob["plink"], None, snps, indiv, prob["f"], n)
    finally:
        os.environ.pop("MIRACULIX_NUM_GPUS")
    try:
        assert dg.num_shards(sh) == 3
        C = dg.dgemm_compressed_main(False, sh, np.asfortranarray(B.T), snps, indiv)
        assert np.abs(C.T - ref).max() <= RTOL * np.abs(ref).max()
    finally:
        dg.free_compressed(sh)
    bed = tmp_path / "x.bed"
    with open(bed, "wb") as fh:
        fh.write(bytes([0x6c, 0x1b, 0x01])); fh.write(prob["plink"].tobytes())
    (tmp_path / "x.bim").write_text("\n".join("1 s 0 0 A B" for _ in range(snps)) + "\n")
    (tmp_path / "x.fam").write_text("\n".join("f i 0 0 0 0" for _ in range(indiv)) + "\n")
    with _single():
        ob, f, s_, i_ = dg.init_compressed_from_bed(str(bed), n)
    try:
        assert (s_, i_) == (snps, indiv)
        C = dg.dgemm_compressed_main(False, ob, np.asfortranarray(B.T), snps, indiv)
        prob_f = dict(prob, f=f)
        reff = o.dgemm_dense(0, prob_f, B, 1)[:, :indiv]
        assert np.abs(C.T - reff).max() <= RTOL * np.abs(reff).max()
    finally:
        dg.free_compressed(ob)


def test_two_copies_that_do_not_fit_keep_one(mx):
    """MXA_SINGLE_ORIENTATION=0 (both copies asked for; one copy is the default since round 5): where the reference's pre-flight stops with "Not enough device
    memory" (cuda_utils.cu:162-185) because the two packed copies do not fit, the object keeps the SNP-major copy alone if that fits and says so on stderr --
    same results (bit-identical on the fp64 MFMA path); a sharded object decides once for all its shards.  The device is filled with a
    ballast tensor so that the case costs megabytes, not hundreds of gigabytes."""
    import torch
    from bench import synth_genotypes_device
    dg = mx.dgemm_compressed
    L = mx.check_library_handle()
    L.mxa_single_orientation.argtypes = [ctypes.c_void_p]
    dev = torch.device("cuda", 0)
    snps, indiv, n = 48_000, 32_000, 8
    plink = synth_genotypes_device(torch, snps, indiv, 5, dev)
    f = mx.read_plink.calc_freq(plink, snps, indiv)
    g = torch.Generator(device=dev); g.manual_seed(6)
    Bn = torch.randn((n, snps), dtype=torch.float64, device=dev, generator=g).t()
    Bt = torch.randn((n, indiv), dtype=torch.float64, device=dev, generator=g).t()
    dg.set_options(use_gpu=True, not_center=False, verbose=0)
    os.environ["MXA_SINGLE_ORIENTATION"] = "0"

    def products(obj):
        out = [dg.dgemm_compressed_main(False, obj, Bn, snps, indiv).clone(), dg.dgemm_compressed_main(True, obj, Bt, snps, indiv).clone()]
        torch.cuda.synchronize()
        return out

    obj = dg.init_compressed(plink, None, snps, indiv, f, n)          # plenty of room: two copies (this also loads every code object before the squeeze)
    assert L.mxa_single_orientation(obj) == 0
    ref = products(obj)
    dg.free_compressed(obj)
    torch.cuda.synchronize(); torch.cuda.empty_cache()
    one_copy = (snps + 256) * ((indiv + 128) // 4)                     # bytes of one packed copy with its tile padding: 388 MB
    free_b, _ = torch.cuda.mem_get_info()
    ballast = torch.empty(free_b - int(1.5 * one_copy), dtype=torch.uint8, device=dev)   # room for one copy and the workspace, not for two
    try:
        obj = dg.init_compressed(plink, None, snps, indiv, f, n)
        try:
            assert L.mxa_single_orientation(obj) == 1
            got = products(obj)
            assert torch.equal(got[0], ref[0]) and torch.equal(got[1], ref[1])
        finally:
            dg.free_compressed(obj)
        os.environ["MIRACULIX_NUM_GPUS"] = "3"
        try:
            obj = dg.init_compressed(plink, None, snps, indiv, f, n)
        finally:
            os.environ.pop("MIRACULIX_NUM_GPUS", None)
        try:
            assert dg.num_shards(obj) == 3 and L.mxa_single_orientation(obj) == 1
            got = products(obj)
            assert torch.equal(got[1], ref[1])                                            # 'T': disjoint row blocks, the same arithmetic per row
            scale = float(ref[0].abs().max())
            assert float((got[0] - ref[0]).abs().max()) <= RTOL * scale                    # 'N': the sum over the SNP blocks is associated differently
        finally:
            dg.free_compressed(obj)
    finally:
        os.environ.pop("MXA_SINGLE_ORIENTATION", None)
        del ballast
        torch.cuda.empty_cache()


def test_bed_staging_under_a_memory_squeeze_keeps_one_copy(mx, tmp_path):
    """the automatic policy on the .bed route (mxa_bed2compressed: raw block + its transpose + two packed copies do not fit, raw block + one packed copy
    do), single-device and sharded: the object keeps one copy and multiplies like the two-copy object created before the squeeze"""
    import torch
    from bench import synth_genotypes_device
    dg = mx.dgemm_compressed
    L = mx.check_library_handle()
    L.mxa_single_orientation.argtypes = [ctypes.c_void_p]
    dev = torch.device("cuda", 0)
    snps, indiv, n = 32_000, 24_000, 8
    plink = synth_genotypes_device(torch, snps, indiv, 9, dev)
    bed = tmp_path / "y.bed"
    with open(bed, "wb") as fh:
        fh.write(bytes([0x6C, 0x1B, 0x01]))
        fh.write(plink.cpu().numpy().tobytes())
    (tmp_path / "y.bim").write_text("\n".join("1 s 0 0 A B" for _ in range(snps)) + "\n")
    (tmp_path / "y.fam").write_text("\n".join("f i 0 0 0 0" for _ in range(indiv)) + "\n")
    del plink
    g = torch.Generator(device=dev); g.manual_seed(10)
    Bn = torch.randn((n, snps), dtype=torch.float64, device=dev, generator=g).t()
    Bt = torch.randn((n, indiv), dtype=torch.float64, device=dev, generator=g).t()
    dg.set_options(use_gpu=True, not_center=False, verbose=0)
    os.environ["MXA_SINGLE_ORIENTATION"] = "0"

    def products(obj):
        out = [dg.dgemm_compressed_main(False, obj, Bn, snps, indiv).clone(), dg.dgemm_compressed_main(True, obj, Bt, snps, indiv).clone()]
        torch.cuda.synchronize()
        return out

    obj, _, _, _ = dg.init_compressed_from_bed(str(bed), n)
    assert L.mxa_single_orientation(obj) == 0
    ref = products(obj)
    dg.free_compressed(obj)
    torch.cuda.synchronize(); torch.cuda.empty_cache()
    one_copy = (snps + 256) * ((indiv + 128) // 4)                     # 194 MB; the raw block is about as big
    free_b, _ = torch.cuda.mem_get_info()
    ballast = torch.empty(free_b - int(2.7 * one_copy), dtype=torch.uint8, device=dev)   # raw + one copy fit (2x), raw + transpose + two copies (4x) do not
    try:
        for shards in (1, 2):
            if shards > 1:
                os.environ["MIRACULIX_NUM_GPUS"] = str(shards)
            try:
                obj, _, _, _ = dg.init_compressed_from_bed(str(bed), n)
            finally:
                os.environ.pop("MIRACULIX_NUM_GPUS", None)
            try:
                assert dg.num_shards(obj) == shards and L.mxa_single_orientation(obj) == 1
                got = products(obj)
                assert torch.equal(got[1], ref[1])
                if shards == 1:
                    assert torch.equal(got[0], ref[0])
                else:
                    assert float((got[0] - ref[0]).abs().max()) <= RTOL * float(ref[0].abs().max())
            finally:
                dg.free_compressed(obj)
    finally:
        os.environ.pop("MXA_SINGLE_ORIENTATION", None)
        del ballast
        torch.cuda.empty_cache()
