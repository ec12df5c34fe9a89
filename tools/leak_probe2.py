#!/usr/bin/env python3
"""does a 2-D device-to-host copy leak per CALL on a long-lived object (main thread)?  400k x 2001, n = 32, 'T' with host C (row-range pipeline)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import miraculix_amd as mx
from bench import synth_genotypes_device
mx.load_shared_library()
dg = mx.dgemm_compressed
dev = torch.device("cuda", 0)
snps, indiv, n = 300_000, 2_001, 32
plink = synth_genotypes_device(torch, snps, indiv, 9, dev)
plink_t = mx.compressed_operations.transpose_genotype_matrix(plink, snps, indiv)
f = mx.read_plink.calc_freq(plink, snps, indiv)
dg.set_options(use_gpu=True, not_center=True, verbose=0)
obj = dg.init_compressed(plink, plink_t, snps, indiv, f, n)
BT = np.asfortranarray(np.random.default_rng(0).standard_normal((indiv, n)))
C = np.zeros((snps + 5, n), order="F")[:snps]     # ldc = snps + 5: strided columns
def free(): torch.cuda.synchronize(); return torch.cuda.mem_get_info()[0]
import ctypes, time
L = mx.check_library_handle()
base = None
t0 = None
fresh = len(sys.argv) > 1 and sys.argv[1] == "fresh"
for it in range(60):
    if fresh:
        C = np.zeros((snps + 5, n), order="F")[:snps]     # a new destination buffer every call, as a binding that allocates its result does
    L.dgemm_compressed(b"T", obj, n, BT.ctypes.data_as(ctypes.c_void_p), indiv, C.ctypes.data_as(ctypes.c_void_p), snps + 5)
    if it == 9: base = free(); t0 = time.perf_counter()
dt = (time.perf_counter() - t0) / 50
print(f"leak probe 2 (MXA_COPY_COLUMNS={os.environ.get('MXA_COPY_COLUMNS','0')}, {'fresh' if fresh else 'same'} destination): drift over 50 calls {(base - free()) / 2**20:.1f} MiB, {dt*1e3:.2f} ms per call")
dg.free_compressed(obj)
