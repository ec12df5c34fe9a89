#!/usr/bin/env python3
"""snp_multiply_gpu through the plain reference ABI (host input, host result): wall time with and without the pipelined
device-to-host copy.  usage: perf_crossprod_host.py k(snps) rows(indiv)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import miraculix_amd as mx
from bench import synth_plink_device

k, rows = int(sys.argv[1]), int(sys.argv[2])
mx.load_shared_library()
X = synth_plink_device(torch, rows, (k + 3) // 4, 7, torch.device("cuda", 0)).cpu().numpy()
M = np.zeros((rows, rows))
for label, env in (("pipelined", None), ("unpipelined", "1"), ("pipelined", None)):
    if env: os.environ["MXA_XPROD_NO_PIPELINE"] = env
    else: os.environ.pop("MXA_XPROD_NO_PIPELINE", None)
    t0 = time.perf_counter()
    mx.crossproduct.snp_crossprod(X, k, rows, is_snpmajor=False, is_plink_format=True, out=M)
    dt = time.perf_counter() - t0
    print(f"snp_multiply_gpu host->host k={k} rows={rows} ({X.nbytes/1e9:.1f} GB in, {M.nbytes/1e9:.1f} GB out) {label}: {dt:.2f} s = {2.0*k*rows*rows/dt*1e-15:.2f} Pop/s PCIe-inclusive", flush=True)
assert M[5, 7] == M[7, 5]
