#!/usr/bin/env python3
"""A/B of one environment knob that the library reads per product, INSIDE one process: blocks of products alternate between the values, so both see the
same box, clock and temperature (separate runs differ by 3-5 % on this pool -- more than most effects worth measuring).
usage: ab_env.py KNOB v0,v1[,v2] snps indiv n [N|T|G] [blocks] [per_block]     (G = gram_matvec)
prints per value: mean / min of the main kernel's device time per product (mxa_profile_get) and of the wall time per product."""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import miraculix_amd as mx
from bench import synth_plink_device

knob, vals = sys.argv[1], sys.argv[2].split(",")
snps, indiv, n = int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
what = sys.argv[6] if len(sys.argv) > 6 else "N"
blocks = int(sys.argv[7]) if len(sys.argv) > 7 else 12
per = int(sys.argv[8]) if len(sys.argv) > 8 else 10
dev = torch.device("cuda", 0)
L = mx.load_shared_library()
plink = synth_plink_device(torch, snps, (indiv + 3) // 4, 42, dev)
f = mx.read_plink.calc_freq(plink, snps, indiv)
dg = mx.dgemm_compressed
dg.set_options(use_gpu=True, not_center=True, verbose=0)
obj = dg.init_compressed(plink, None, snps, indiv, f, n)
del plink
g = torch.Generator(device=dev); g.manual_seed(1)
trans = what == "T"
k, m = (indiv, snps) if trans else (snps, indiv)
if what == "G":
    k = m = indiv
B = torch.randn((n, k), dtype=torch.float64, device=dev, generator=g).t()
C = torch.zeros((n, m), dtype=torch.float64, device=dev).t()

def product():
    if what == "G":
        dg.gram_matvec(obj, B, snps, indiv, out=C)
    else:
        dg.dgemm_compressed_main(trans, obj, B, snps, indiv, out=C)

t_w = time.perf_counter()
while time.perf_counter() - t_w < 1.0:
    product(); torch.cuda.synchronize()
kern = {v: [] for v in vals}; wall = {v: [] for v in vals}
for b in range(blocks):
    for v in (vals if b % 2 == 0 else vals[::-1]):
        os.environ[knob] = v
        product(); torch.cuda.synchronize()
        L.mxa_profile_reset()
        t0 = time.perf_counter()
        for _ in range(per):
            product()
        torch.cuda.synchronize()
        wall[v].append((time.perf_counter() - t0) / per * 1e3)
        la, ms = ctypes.c_int(0), ctypes.c_double(0)
        L.mxa_profile_get(ctypes.byref(la), ctypes.byref(ms))
        kern[v].append(ms.value / max(1, la.value))
for v in vals:
    kk, ww = kern[v], wall[v]
    print(f"{knob}={v}: {what} {snps} x {indiv} n={n}: main kernel mean {sum(kk)/len(kk):.4f} ms (min {min(kk):.4f}), wall per product mean {sum(ww)/len(ww):.4f} ms (min {min(ww):.4f}); path {dg.last_path()}", flush=True)
dg.free_compressed(obj)
