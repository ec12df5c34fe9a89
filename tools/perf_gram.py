#!/usr/bin/env python3
"""per-call wall time of the CG step: mxa_gram_matvec vs the 'T' + 'N' pair, n = 1 (config-5 shard by default)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import miraculix_amd as mx
from bench import synth_plink_device

snps, indiv = int(sys.argv[1]), int(sys.argv[2])
n = int(sys.argv[3]) if len(sys.argv) > 3 else 1
dev = torch.device("cuda", 0)
L = mx.load_shared_library()
plink = synth_plink_device(torch, snps, (indiv + 3) // 4, 42, dev)
plink_t = mx.compressed_operations.transpose_genotype_matrix(plink, snps, indiv)
f = mx.read_plink.calc_freq(plink, snps, indiv)
dg = mx.dgemm_compressed
dg.set_options(use_gpu=True, not_center=False, verbose=0)
obj = dg.init_compressed(plink, plink_t, snps, indiv, f, n)
del plink, plink_t
V = torch.randn((n, indiv), dtype=torch.float64, device=dev).t()
T = torch.zeros((n, snps), dtype=torch.float64, device=dev).t()
O = torch.zeros((n, indiv), dtype=torch.float64, device=dev).t()
for name, fn in (("T+N pair", lambda: (dg.dgemm_compressed_main(True, obj, V, snps, indiv, out=T), dg.dgemm_compressed_main(False, obj, T, snps, indiv, out=O))),
                 ("gram_matvec", lambda: dg.gram_matvec(obj, V, snps, indiv, out=O))):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 20
    print(f"{name}: {dt*1e3:.3f} ms per G*v ({snps} x {indiv}, n={n}) = {4.0*snps*indiv*n/dt*1e-12:.2f} TFLOP/s", flush=True)
dg.free_compressed(obj)
