"""Host-side mirror of `miraculix.read_plink` (src/bindings/Julia/read_plink.jl:161-222): .bed reader (3-byte magic
6c 1b 01, SNP-major) and popcount allele frequencies computed on the device."""
import numpy as np

from . import lib as _lib

BED_MAGIC = bytes([0x6C, 0x1B, 0x01])


def read_bed(path, snps=None, indiv=None):
    """Returns (plink [snps x ceil(indiv/4)] uint8 without the header, snps, indiv).  If a .bim/.fam pair sits next to the
    .bed the dimensions are taken from their line counts (read_plink.jl:165-171), else snps and indiv must be given."""
    base = path[:-4] if path.endswith(".bed") else path
    if snps is None or indiv is None:
        with open(base + ".bim") as fh:
            snps = sum(1 for _ in fh)
        with open(base + ".fam") as fh:
            indiv = sum(1 for _ in fh)
    with open(base + ".bed", "rb") as fh:
        magic = fh.read(3)
        if magic != BED_MAGIC:
            raise ValueError("not a SNP-major PLINK .bed file (magic bytes 6c 1b 01 expected)")
        data = np.frombuffer(fh.read(), dtype=np.uint8)
    bps = (indiv + 3) // 4
    if data.size != snps * bps:
        raise ValueError(f".bed payload has {data.size} bytes, expected {snps * bps}")
    return data.reshape(snps, bps).copy(), snps, indiv


def write_bed(path, plink):
    with open(path, "wb") as fh:
        fh.write(BED_MAGIC)
        fh.write(np.ascontiguousarray(plink, dtype=np.uint8).tobytes())


def calc_freq(plink, snps, indiv):
    """f_s = popcount(row s) / (2 indiv) (read_plink.jl:199-203), on the device"""
    L = _lib.check_library_handle()
    if _lib.is_torch_tensor(plink):
        import torch
        f = torch.zeros(snps, dtype=torch.float64, device=plink.device)
    else:
        plink = np.ascontiguousarray(plink, dtype=np.uint8)
        f = np.zeros(snps, dtype=np.float64)
    rc = L.mxa_allele_freq(_lib.ptr(plink), int(snps), int(indiv), _lib.ptr(f))
    if rc != 0:
        raise RuntimeError("mxa_allele_freq failed: " + _lib.last_error()[1])
    return f
