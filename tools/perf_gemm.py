#!/usr/bin/env python3
"""kernel-level perf of dgemm_compressed on synthetic device data: prints avg k_gemm launch ms / TFLOP/s for 'N' and 'T'.
usage: perf_gemm.py snps indiv n reps   (env MXA_GEMM_MODE selects the conversion variant)"""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import miraculix_amd as mx
from bench import synth_plink_device

snps, indiv, n, reps = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
centered = int(os.environ.get("CENTERED", "0"))
dev = torch.device("cuda", 0)
L = mx.load_shared_library()
plink = synth_plink_device(torch, snps, (indiv + 3) // 4, 42, dev)
plink_t = mx.compressed_operations.transpose_genotype_matrix(plink, snps, indiv)
f = mx.read_plink.calc_freq(plink, snps, indiv)
dg = mx.dgemm_compressed
dg.set_options(use_gpu=True, not_center=not centered, verbose=0)
obj = dg.init_compressed(plink, plink_t, snps, indiv, f, n)
del plink, plink_t
g = torch.Generator(device=dev); g.manual_seed(1)
for trans in (False, True):
    k = indiv if trans else snps
    m = snps if trans else indiv
    B = torch.randn((n, k), dtype=torch.float64, device=dev, generator=g).t()
    C = torch.zeros((n, m), dtype=torch.float64, device=dev).t()
    if os.environ.get("TINY_ENTRY"):   # one entry 150 binades below its column's maximum: the exactness verdict of the guarded int8 route declines (fp64 chains)
        B[3, n - 1] = 1e-45
    # warm-up by time, not by count: the clock of an idle GPU ramps over the first ~0.3 s of work (rocprofv3 GRBM_GUI_ACTIVE: 2.07 -> 2.26 GHz over
    # six 3 ms launches), which used to be charged to whichever product ran first
    t_w = time.perf_counter()
    while True:
        dg.dgemm_compressed_main(trans, obj, B, snps, indiv, out=C)
        torch.cuda.synchronize()
        if time.perf_counter() - t_w > float(os.environ.get("WARM_S", "0.5")):
            break
    L.mxa_profile_reset()
    t0 = time.perf_counter()
    for _ in range(reps):
        dg.dgemm_compressed_main(trans, obj, B, snps, indiv, out=C)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / reps
    la, ms = ctypes.c_int(0), ctypes.c_double(0)
    L.mxa_profile_get(ctypes.byref(la), ctypes.byref(ms))
    avg = ms.value / la.value
    fl = 2.0 * snps * indiv * n
    gm, gk, gn, gs, ga, gc = ctypes.c_long(), ctypes.c_long(), ctypes.c_int(), ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    L.mxa_last_geometry(ctypes.byref(gm), ctypes.byref(gk), ctypes.byref(gn), ctypes.byref(gs), ctypes.byref(ga), ctypes.byref(gc))
    print(f"path={dg.last_path()} tile=({ga.value},{gc.value}) splits={gs.value} mode={os.environ.get('MXA_GEMM_MODE','0')} {'T' if trans else 'N'} snps={snps} indiv={indiv} n={n}: kernel {avg:.3f} ms = {fl/avg*1e-9:.2f} TFLOP/s; call wall {wall*1e3:.3f} ms = {fl/wall*1e-12:.2f} TFLOP/s", flush=True)
dg.free_compressed(obj)
