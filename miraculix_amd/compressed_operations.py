"""Host-side mirror of `miraculix.compressed_operations` (src/bindings/Julia/compressed_operations.jl):
2-bit transpose on the device (transpose_genotype_matrix, :45-66) and decompression for checks (:80-110)."""
import numpy as np

from . import lib as _lib


def transpose_genotype_matrix(plink, snps, indiv):
    """plink: `snps` rows of ceil(indiv/4) bytes -> `indiv` rows of ceil(snps/4) bytes (padding bits zero)."""
    L = _lib.check_library_handle()
    if _lib.is_torch_tensor(plink):
        import torch
        out = torch.zeros((indiv, (snps + 3) // 4), dtype=torch.uint8, device=plink.device)
    else:
        plink = np.ascontiguousarray(plink, dtype=np.uint8)
        out = np.zeros((indiv, (snps + 3) // 4), dtype=np.uint8)
    rc = L.mxa_transpose_2bit(_lib.ptr(plink), int(snps), int(indiv), _lib.ptr(out))
    if rc != 0:
        raise RuntimeError("mxa_transpose_2bit failed: " + _lib.last_error()[1])
    return out


def decompress_genotype_matrix(plink, rows, cols):
    """rows x cols int8 matrix of allele counts (PLINK code -> max(code-1,0)); host-side helper for tests"""
    p = np.asarray(plink, dtype=np.uint8).reshape(rows, -1)
    codes = np.stack([(p >> (2 * q)) & 3 for q in range(4)], axis=2).reshape(rows, -1)[:, :cols]
    return np.maximum(codes.astype(np.int8) - 1, 0)
