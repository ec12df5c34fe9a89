#!/usr/bin/env python3
"""wall time of EVERY device-result call of snp_multiply_gpu (config 3 by default), one line per call, with the library's phase clock under PRINT_LEVEL=1:
where a slow call spends its time (calls of 982 / 1154 / 1103 ms beside 783 with identical kernel times).
usage: perf_crossprod_calls.py k(snps) rows(indiv) calls"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import miraculix_amd as mx
from bench import synth_plink_device
k, rows, calls = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
dev = torch.device("cuda", 0)
mx.load_shared_library()
X = synth_plink_device(torch, rows, (k + 3) // 4, 7, dev)
M = torch.zeros((rows, rows), dtype=torch.float64, device=dev)
for c in range(calls):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    mx.crossproduct.snp_crossprod(X, k, rows, is_snpmajor=False, is_plink_format=True, out=M)
    torch.cuda.synchronize()
    print(f"== call {c}: wall {1e3 * (time.perf_counter() - t0):.1f} ms", flush=True)
