"""writes a synthetic PLINK data set <base>.bed/.bim/.fam/.freq (no missings; p_s ~ U(0.1, 0.6), g ~ Binomial(2, p_s)) for the reference's Fortran programs:
python tools/make_bed_dataset.py <base> <snps> <indiv>"""
import sys

import numpy as np

base, snps, indiv = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
rng = np.random.default_rng(42)
bps = (indiv + 3) // 4
code = np.array([0, 2, 3], dtype=np.uint8)          # PLINK: 0 -> 00, 1 -> 10, 2 -> 11
with open(base + ".bed", "wb") as bed, open(base + ".freq", "w") as fq:
    bed.write(bytes([0x6C, 0x1B, 0x01]))
    for s0 in range(0, snps, 4096):
        ns = min(4096, snps - s0)
        p = rng.uniform(0.1, 0.6, size=(ns, 1))
        g = (rng.random((ns, indiv)) < p).astype(np.uint8) + (rng.random((ns, indiv)) < p).astype(np.uint8)
        c = np.zeros((ns, bps * 4), dtype=np.uint8)
        c[:, :indiv] = code[g]
        c = c.reshape(ns, bps, 4)
        bed.write((c[:, :, 0] | (c[:, :, 1] << 2) | (c[:, :, 2] << 4) | (c[:, :, 3] << 6)).astype(np.uint8).tobytes())
        f = g.mean(axis=1) / 2.0
        fq.write("".join(f"{s0 + i + 1} {f[i]:.17g}\n" for i in range(ns)))
with open(base + ".bim", "w") as fh:
    fh.write("".join(f"1 snp{i} 0 {i} A B\n" for i in range(snps)))
with open(base + ".fam", "w") as fh:
    fh.write("".join(f"f{i} i{i} 0 0 0 -9\n" for i in range(indiv)))
