"""GRM / LD with the element-wise post-processing of the reference's binding (src/bindings/Julia/crossproduct.jl:83-152: two BLAS.ger!, the
affine shift and the scaling for the GRM; syr! and the division by sigma sigma^T for LD) FUSED into the crossproduct epilogue: the column sums and
the diagonal of M = X X^T come from the staged 2-bit matrix, not from three more passes over the result.  Checked against a dense numpy
restatement of those lines (the reference tests' own oracle: tests/crossproduct/test_grm.jl:114-141, test_ld.jl:68-80, tolerance stated here:
1e-12 relative, far inside the reference's 1e-4 / 0.1) and BIT FOR BIT against the unfused kernels (MXA_XPROD_FUSED_POST=0), for both engines,
host and device results, ragged sizes, PLINK and raw 2-bit input, and through the slab pipeline of a host result."""
import numpy as np
import pytest

from _util import make_problem, pack_plink

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mx():
    import miraculix_amd as m
    m.load_shared_library()
    return m


def _grm_ref(Z, f, do_scale):
    """crossproduct.jl:94-107, literally: M = Z Z^T; ger, ger, affine shift, scaling"""
    M = Z @ Z.T
    n = M.shape[0]
    cs = M.sum(axis=0)
    M = M - np.outer(cs, np.ones(n)) / n - np.outer(np.ones(n), cs) / n + cs.sum() / n ** 2
    return M / (2 * np.sum(f * (1 - f))) if do_scale else M


def _ld_ref(Z, f, indiv):
    """crossproduct.jl:137-149"""
    M = Z.T @ Z - 4.0 * indiv * np.outer(f, f)
    s = np.sqrt(np.diag(M))
    return M / s[:, None] / s[None, :]


@pytest.mark.parametrize("snps,indiv", [(3000, 400), (777, 515), (130, 1031), (5000, 257)])
@pytest.mark.parametrize("engine", ["f4", "i8"])
def test_fused_grm_and_ld_equal_the_unfused_kernels_and_the_dense_restatement(mx, monkeypatch, snps, indiv, engine):
    monkeypatch.setenv("MXA_XPROD_ENGINE", engine)
    prob = make_problem(snps, indiv, 1, seed=snps + indiv)
    Z = prob["Z"].astype(np.float64)          # indiv x snps, values 0 / 1 / 2 (no missings)
    f = prob["f"]
    cp = mx.crossproduct
    for do_scale in (True, False):
        monkeypatch.setenv("MXA_XPROD_FUSED_POST", "1")
        G = cp.grm(prob["plink_t"], snps, indiv, is_plink_format=True, do_scale=do_scale, allele_freq=f if do_scale else None)
        monkeypatch.setenv("MXA_XPROD_FUSED_POST", "0")
        G0 = cp.grm(prob["plink_t"], snps, indiv, is_plink_format=True, do_scale=do_scale, allele_freq=f if do_scale else None)
        assert np.array_equal(G, G0)
        ref = _grm_ref(Z, f, do_scale)
        assert np.abs(G - ref).max() <= 1e-12 * np.abs(ref).max()
    monkeypatch.setenv("MXA_XPROD_FUSED_POST", "1")
    R = cp.ld(prob["plink"], snps, indiv, is_plink_format=True, allele_freq=f)
    monkeypatch.setenv("MXA_XPROD_FUSED_POST", "0")
    R0 = cp.ld(prob["plink"], snps, indiv, is_plink_format=True, allele_freq=f)
    assert np.array_equal(R, R0, equal_nan=True)
    ref = _ld_ref(Z, f, indiv)
    ok = np.isfinite(ref)                     # a monomorphic SNP has sigma = 0 in the reference too
    assert np.array_equal(np.isfinite(R), ok)
    assert np.abs(R[ok] - ref[ok]).max() <= 1e-11


def test_fused_grm_device_resident_and_through_the_host_slab_pipeline(mx, monkeypatch):
    import torch
    dev = torch.device("cuda", 0)
    snps, indiv = 2100, 1290
    prob = make_problem(snps, indiv, 1, seed=8)
    f = prob["f"]
    cp = mx.crossproduct
    ref = _grm_ref(prob["Z"].astype(np.float64), f, True)
    Gd = cp.grm(torch.from_numpy(prob["plink_t"]).to(dev), snps, indiv, is_plink_format=True, do_scale=True, allele_freq=torch.from_numpy(f).to(dev))
    assert Gd.is_cuda
    monkeypatch.setenv("MXA_XPROD_SLAB_MB", "3")            # one tile row per chunk: six chunks, copied out while the next one computes
    Gh = cp.grm(prob["plink_t"], snps, indiv, is_plink_format=True, do_scale=True, allele_freq=f)
    monkeypatch.setenv("MXA_XPROD_NO_PIPELINE", "1")
    Gn = cp.grm(prob["plink_t"], snps, indiv, is_plink_format=True, do_scale=True, allele_freq=f)
    assert np.array_equal(Gd.cpu().numpy(), Gh) and np.array_equal(Gh, Gn)
    assert np.abs(Gh - ref).max() <= 1e-12 * np.abs(ref).max()
    Rh = cp.ld(prob["plink"], snps, indiv, is_plink_format=True, allele_freq=f)
    monkeypatch.delenv("MXA_XPROD_NO_PIPELINE")
    Rp = cp.ld(prob["plink"], snps, indiv, is_plink_format=True, allele_freq=f)
    assert np.array_equal(Rh, Rp, equal_nan=True)


def test_fused_grm_raw_two_bit_values_up_to_three(mx, monkeypatch):
    """is_plink_format = False: the packed fields are the values themselves, 3 included (the column sums and the diagonal must count 3 and 9)"""
    rng = np.random.default_rng(11)
    rows, k = 389, 1203
    V = rng.integers(0, 4, size=(rows, k)).astype(np.uint8)
    Vp = np.zeros((rows, (k + 3) // 4 * 4), dtype=np.uint8); Vp[:, :k] = V
    X = np.ascontiguousarray((Vp[:, 0::4] | (Vp[:, 1::4] << 2) | (Vp[:, 2::4] << 4) | (Vp[:, 3::4] << 6)).astype(np.uint8))
    Zf = V.astype(np.float64)
    f = rng.uniform(0.1, 0.5, size=k)
    cp = mx.crossproduct
    G = cp.grm(X, k, rows, is_plink_format=False, do_scale=True, allele_freq=f)
    monkeypatch.setenv("MXA_XPROD_FUSED_POST", "0")
    G0 = cp.grm(X, k, rows, is_plink_format=False, do_scale=True, allele_freq=f)
    assert np.array_equal(G, G0)
    ref = _grm_ref(Zf, f, True)
    assert np.abs(G - ref).max() <= 1e-12 * np.abs(ref).max()
