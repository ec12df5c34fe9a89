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


def grm(plink_transposed, snps, indiv, is_plink_format=False, do_scale=True, allele_freq=None):
    """crossproduct.jl:83-110; maths docs/grm.md:5-12.  G = P Z Z^T P^T / (2 sum f(1-f)) via rank-1 updates of M = Z Z^T."""
    if do_scale and (allele_freq is None or len(allele_freq) != snps):
        raise ValueError(f"Allele frequencies need to be equal to length of SNPs {snps}.")
    M = np.asarray(snp_crossprod(plink_transposed, snps, indiv, is_snpmajor=False, is_plink_format=is_plink_format))
    col_sum = M.sum(axis=0)
    M = M - np.outer(col_sum, np.ones(indiv)) / indiv - np.outer(np.ones(indiv), col_sum) / indiv
    M += col_sum.sum() / indiv**2
    if do_scale:
        f = np.asarray(allele_freq, dtype=np.float64)
        M /= 2.0 * np.sum(f * (1.0 - f))
    return M


def ld(plink, snps, indiv, is_plink_format=False, allele_freq=None):
    """crossproduct.jl:128-152: R^2-type LD statistic from the SNP x SNP crossproduct."""
    if allele_freq is None or len(allele_freq) != snps:
        raise ValueError(f"Allele frequencies need to be equal to length of SNPs {snps}.")
    M = np.asarray(snp_crossprod(plink, snps, indiv, is_snpmajor=True, is_plink_format=is_plink_format)).copy()
    f = np.asarray(allele_freq, dtype=np.float64)
    M -= 4.0 * indiv * np.outer(f, f)
    sigma = np.sqrt(np.diag(M))
    M /= sigma[:, None]
    M /= sigma[None, :]
    return M
