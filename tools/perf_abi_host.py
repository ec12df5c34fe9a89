#!/usr/bin/env python3
"""PCIe-inclusive rate of the plain reference ABI: host B and host C (numpy), as the reference's bindings call it.
Never used as bench.py's `value`."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import miraculix_amd as mx
from bench import synth_plink_device
snps, indiv, n = 400_000, 50_000, 32
dev = torch.device("cuda", 0)
mx.load_shared_library()
plink = synth_plink_device(torch, snps, (indiv + 3) // 4, 42, dev)
plink_t = mx.compressed_operations.transpose_genotype_matrix(plink, snps, indiv)
f = mx.read_plink.calc_freq(plink, snps, indiv)
dg = mx.dgemm_compressed
dg.set_options(use_gpu=True, not_center=True, verbose=0)
t0 = time.perf_counter(); ph, pth = plink.cpu().numpy(), plink_t.cpu().numpy(); fh = f.cpu().numpy()
t0 = time.perf_counter()
obj = dg.init_compressed(ph, pth, snps, indiv, fh, n)       # host staging: 2 x 5 GB over PCIe
t_stage = time.perf_counter() - t0
print(f"plink2compressed from HOST buffers ({(ph.nbytes + pth.nbytes)/1e9:.1f} GB): {t_stage:.2f} s = {(ph.nbytes + pth.nbytes)/t_stage/1e9:.1f} GB/s")
rng = np.random.default_rng(0)
for trans in (False, True):
    k = indiv if trans else snps
    m = snps if trans else indiv
    B = np.asfortranarray(rng.standard_normal((k, n)))
    C = np.zeros((m, n), order="F")
    dg.dgemm_compressed_main(trans, obj, B, snps, indiv, out=C)
    t0 = time.perf_counter()
    for _ in range(3):
        dg.dgemm_compressed_main(trans, obj, B, snps, indiv, out=C)
    dt = (time.perf_counter() - t0) / 3
    print(f"dgemm_compressed '{'T' if trans else 'N'}' {snps} x {indiv} x {n}, HOST B ({B.nbytes/1e6:.0f} MB) and HOST C ({C.nbytes/1e6:.0f} MB): {dt*1e3:.1f} ms = {2.0*snps*indiv*n/dt*1e-12:.1f} TFLOP/s PCIe-inclusive")
dg.free_compressed(obj)
