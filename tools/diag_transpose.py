#!/usr/bin/env python3
"""round-trip check of mxa_transpose_2bit at large shapes; prints where the round trip first differs"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import miraculix_amd as mx
from bench import synth_plink_device
mx.load_shared_library()
dev = torch.device("cuda", 0)
for rows, cols in [(int(a), int(b)) for a, b in (s.split("x") for s in sys.argv[1:])]:
    P = synth_plink_device(torch, rows, (cols + 3) // 4, 3, dev)
    if cols % 4:
        P[:, -1] &= (1 << (2 * (cols % 4))) - 1
    T = mx.compressed_operations.transpose_genotype_matrix(P, rows, cols)
    back = mx.compressed_operations.transpose_genotype_matrix(T, cols, rows)
    ok = torch.equal(back, P)
    msg = ""
    if not ok:
        bad = (back != P).any(dim=1).nonzero().flatten()
        msg = f" first bad row {int(bad[0])}, last {int(bad[-1])}, count {bad.numel()}"
        # is T or back wrong?  popcount of rows of P vs columns in T is hard; check T's nonzero row-dword range instead
        nzc = (T != 0).any(dim=0).nonzero().flatten()
        msg += f"; T nonzero byte-columns {int(nzc[0])}..{int(nzc[-1])} of {T.shape[1]}"
    print(f"transpose {rows} x {cols}: round trip {'ok' if ok else 'FAILED'}{msg}", flush=True)
    del P, T, back
    torch.cuda.empty_cache()
