"""Host B / C through the plain ABI at sizes where the transfers are pipelined behind the product (mxa_api.cpp:
gemm_host_pipelined): K-range launches with per-range exponents for a big B ('N'), row-range launches for a big C ('T').  The results
must be bit-identical to the device-resident one-launch path (same split-K boundaries, same summation order, exact power-of-two
scalings), honour padded leading dimensions, and agree with the oracle on sampled rows."""
import ctypes

import numpy as np
import pytest

from _util import Oracle

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def setup():
    import torch
    import miraculix_amd as mx
    from bench import synth_genotypes_device
    mx.load_shared_library()
    dev = torch.device("cuda", 0)
    snps, indiv = 140_003, 3_001            # B of 'N' and C of 'T': 140 003 x 32 doubles = 35.8 MB (> the 32 MB threshold)
    plink = synth_genotypes_device(torch, snps, indiv, 5, dev)
    plink_t = mx.compressed_operations.transpose_genotype_matrix(plink, snps, indiv)
    f = mx.read_plink.calc_freq(plink, snps, indiv)
    dg = mx.dgemm_compressed
    dg.set_options(use_gpu=True, not_center=False, verbose=0)
    obj = dg.init_compressed(plink, plink_t, snps, indiv, f, 40)
    yield dict(torch=torch, mx=mx, dg=dg, obj=obj, snps=snps, indiv=indiv, plink=plink, plink_t=plink_t, f=f, dev=dev)
    dg.free_compressed(obj)


@pytest.mark.parametrize("n", [32, 40, 34])
@pytest.mark.parametrize("centered", [0, 1])
def test_pipelined_host_path_is_bitwise_the_device_path(setup, n, centered):
    torch, mx, dg, obj, snps, indiv = (setup[k] for k in ("torch", "mx", "dg", "obj", "snps", "indiv"))
    L = mx.check_library_handle()
    dg.set_options(use_gpu=True, not_center=not centered, verbose=0)
    rng = np.random.default_rng(n)
    for trans in (0, 1):
        k = indiv if trans else snps
        m = snps if trans else indiv
        ldb, ldc = k + 3, m + 5
        B = np.zeros((n, ldb)); B[:, :k] = rng.standard_normal((n, k)) * 10.0 ** rng.uniform(-3, 3, size=(n, 1)); B[:, k:] = 1e300
        C = np.full((n, ldc), -777.0)
        L.dgemm_compressed(b"T" if trans else b"N", obj, n, B.ctypes.data_as(ctypes.c_void_p), ldb, C.ctypes.data_as(ctypes.c_void_p), ldc)
        assert L.mxa_last_error() == 0
        Bd = torch.from_numpy(B).to(setup["dev"])
        Cd = torch.full((n, ldc), -777.0, dtype=torch.float64, device=setup["dev"])
        L.dgemm_compressed(b"T" if trans else b"N", obj, n, ctypes.c_void_p(Bd.data_ptr()), ldb, ctypes.c_void_p(Cd.data_ptr()), ldc)
        assert L.mxa_last_error() == 0
        assert np.array_equal(C, Cd.cpu().numpy()), (trans, n, centered)
        assert np.all(C[:, m:] == 0.0)
        # repeatable
        C2 = np.full((n, ldc), -1.0)
        L.dgemm_compressed(b"T" if trans else b"N", obj, n, B.ctypes.data_as(ctypes.c_void_p), ldb, C2.ctypes.data_as(ctypes.c_void_p), ldc)
        assert np.array_equal(C, C2)


def test_pipelined_host_path_vs_oracle_sampled(setup):
    torch, dg, obj, snps, indiv = (setup[k] for k in ("torch", "dg", "obj", "snps", "indiv"))
    o = Oracle()
    dg.set_options(use_gpu=True, not_center=False, verbose=0)
    n = 32
    rng = np.random.default_rng(1)
    f = setup["f"].cpu().numpy()
    cols = [0, 15, 31]
    # 'N': 48 sampled individuals
    B = rng.standard_normal((n, snps))
    C = dg.dgemm_compressed_main(False, obj, np.asfortranarray(B.T), snps, indiv)           # indiv x n, host operands
    ii = np.sort(rng.choice(indiv, 48, replace=False))
    rows = setup["plink_t"][torch.from_numpy(ii).to(setup["dev"])].cpu().numpy()
    sub = o.transpose_2bit(np.ascontiguousarray(rows), 48, snps)
    ref = o.dgemm_dense(0, dict(snps=snps, indiv=48, plink=sub, plink_t=rows, f=f), np.ascontiguousarray(B[cols]), 1)
    assert np.abs(C[ii][:, cols].T - ref).max() <= 1e-11 * np.abs(ref).max()
    # 'T': 48 sampled SNPs spread over all four row ranges
    B = rng.standard_normal((n, indiv))
    C = dg.dgemm_compressed_main(True, obj, np.asfortranarray(B.T), snps, indiv)            # snps x n
    ss = np.sort(np.concatenate([rng.choice(snps, 44, replace=False), [0, 35327, 35328, snps - 1]]))
    srows = setup["plink"][torch.from_numpy(ss).to(setup["dev"])].cpu().numpy()
    ref = o.dgemm_dense(1, dict(snps=len(ss), indiv=indiv, plink=np.ascontiguousarray(srows), plink_t=None, f=np.ascontiguousarray(f[ss])), np.ascontiguousarray(B[cols]), 1)
    assert np.abs(C[ss][:, cols].T - ref).max() <= 1e-11 * np.abs(ref).max()
