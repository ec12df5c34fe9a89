"""Host-side mirror of the reference's Julia module `miraculix.crossproduct`
(src/bindings/Julia/crossproduct.jl:44-152): snp_crossprod, grm, ld over the C symbol snp_multiply_gpu."""
import numpy as np

from . import lib as _lib


def snp_crossprod(plink, snps, indiv, is_snpmajor, is_plink_format=False, out=None):
    """crossproduct.jl:44-64.  plink: 2-bit matrix, one row per output index: for is_snpmajor=False `indiv` rows of
    ceil(snps/4) bytes (result indiv x indiv), for is_snpmajor=True `snps` rows of ceil(indiv/4) bytes (result snps x snps).
    numpy uint8 (host) or torch uint8 tensor (host/device).  Returns the full symmetric Float64 matrix (numpy, or a torch
    tensor on the input's device)."""
    if is_snpmajor:
        nrow, ncol = indiv, snps
    else:
        nrow, ncol = snps, indiv
    nbytes = int(np.prod(plink.shape))
    if nbytes != ncol * ((nrow + 3) // 4):
        raise ValueError(f"Matrix has wrong dimensions: {tuple(plink.shape)}")
    L = _lib.check_library_handle()
    if out is not None:
        M = out
    elif _lib.is_torch_tensor(plink):
        import torch
        M = torch.zeros((ncol, ncol), dtype=torch.float64, device=plink.device)
    else:
        M = np.zeros((ncol, ncol), dtype=np.float64)
    rc = L.snp_multiply_gpu(_lib.ptr(plink), int(nrow), int(ncol), _lib.ptr(M), bool(is_plink_format))
    if rc != 0:
        raise RuntimeError("snp_multiply_gpu failed: " + _lib.last_error()[1])
    return M


def snp_crossprod_panel(plink, inner, n_out, col_begin, col_end, upper_only=False, is_plink_format=False, out=None):
    """Additive (C entry mxa_snp_multiply_panel): columns [col_begin, col_end) of the symmetric n_out x n_out crossproduct of the
    2-bit matrix `plink` (n_out rows of ceil(inner/4) bytes).  Returns the panel as a tensor/array P of shape
    (col_end - col_begin, n_out) with P[c, r] = M[r, col_begin + c] -- the contiguous slab of the column-major result.
    upper_only: only rows [0, col_end) are computed (zeros below)."""
    if int(np.prod(plink.shape)) != n_out * ((inner + 3) // 4):
        raise ValueError(f"Matrix has wrong dimensions: {tuple(plink.shape)}")
    w = col_end - col_begin
    L = _lib.check_library_handle()
    if out is not None:
        P = out
    elif _lib.is_torch_tensor(plink):
        import torch
        P = torch.zeros((w, n_out), dtype=torch.float64, device=plink.device)
    else:
        P = np.zeros((w, n_out), dtype=np.float64)
    rc = L.mxa_snp_multiply_panel(_lib.ptr(plink), int(inner), int(n_out), int(col_begin), int(col_end), int(bool(upper_only)), _lib.ptr(P), int(n_out),
                                  int(bool(is_plink_format)))
    if rc != 0:
        raise RuntimeError("mxa_snp_multiply_panel failed: " + _lib.last_error()[1])
    return P


def _result_like(plink, n):
    if _lib.is_torch_tensor(plink):
        import torch
        return torch.zeros((n, n), dtype=torch.float64, device=plink.device)
    return np.zeros((n, n), dtype=np.float64)


def grm(plink_transposed, snps, indiv, is_plink_format=False, do_scale=True, allele_freq=None):
    """crossproduct.jl:83-110; maths docs/grm.md:5-12.  G = P Z Z^T P^T / (2 sum f(1-f)): the crossproduct and the rank-1
    centring / scaling all run on the device (C entry mxa_grm); only the finished G crosses PCIe when inputs are host arrays."""
    if do_scale and (allele_freq is None or len(allele_freq) != snps):
        raise ValueError(f"Allele frequencies need to be equal to length of SNPs {snps}.")
    if int(np.prod(plink_transposed.shape)) != indiv * ((snps + 3) // 4):
        raise ValueError(f"Matrix has wrong dimensions: {tuple(plink_transposed.shape)}")
    L = _lib.check_library_handle()
    G = _result_like(plink_transposed, indiv)
    f = allele_freq
    if f is not None and not _lib.is_torch_tensor(f):
        f = np.ascontiguousarray(f, dtype=np.float64)
    rc = L.mxa_grm(_lib.ptr(plink_transposed), int(snps), int(indiv), _lib.ptr(G), int(bool(is_plink_format)), int(bool(do_scale)), _lib.ptr(f))
    if rc != 0:
        raise RuntimeError("mxa_grm failed: " + _lib.last_error()[1])
    return G


def ld(plink, snps, indiv, is_plink_format=False, allele_freq=None):
    """crossproduct.jl:128-152: LD correlation from the SNP x SNP crossproduct, on the device (C entry mxa_ld)."""
    if allele_freq is None or len(allele_freq) != snps:
        raise ValueError(f"Allele frequencies need to be equal to length of SNPs {snps}.")
    if int(np.prod(plink.shape)) != snps * ((indiv + 3) // 4):
        raise ValueError(f"Matrix has wrong dimensions: {tuple(plink.shape)}")
    L = _lib.check_library_handle()
    R = _result_like(plink, snps)
    f = allele_freq if _lib.is_torch_tensor(allele_freq) else np.ascontiguousarray(allele_freq, dtype=np.float64)
    rc = L.mxa_ld(_lib.ptr(plink), int(snps), int(indiv), _lib.ptr(R), int(bool(is_plink_format)), _lib.ptr(f))
    if rc != 0:
        raise RuntimeError("mxa_ld failed: " + _lib.last_error()[1])
    return R
