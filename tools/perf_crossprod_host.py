#!/usr/bin/env python3
"""snp_multiply_gpu through the plain reference ABI (host input, host result) the way a Julia caller uses it: the result is a FRESH
numpy.zeros array on every call (src/bindings/Julia/crossproduct.jl:56 `M = zeros(...)`), so none of its pages exist when the call starts.
A/B: background population of the destination pages (mxa_hostmem.h, default) against none (MXA_PREFAULT_THREADS=0), huge-page hint on / off,
then a call into the already-touched array of the previous call, and the unpipelined path.  PRINT_LEVEL=1 prints where each copier's time went.
usage: perf_crossprod_host.py k(snps) rows(indiv) [repeats]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import miraculix_amd as mx
from bench import synth_plink_device

k, rows = int(sys.argv[1]), int(sys.argv[2])
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 2
mx.load_shared_library()
X = synth_plink_device(torch, rows, (k + 3) // 4, 7, torch.device("cuda", 0)).cpu().numpy()
M = None


def run(label, env, fresh):
    global M
    for key in ("MXA_PREFAULT_THREADS", "MXA_HOST_HUGEPAGE", "MXA_XPROD_NO_PIPELINE"):
        os.environ.pop(key, None)
    os.environ.update(env)
    if fresh or M is None:
        M = None
        M = np.zeros((rows, rows))          # calloc'ed: untouched pages
    t0 = time.perf_counter()
    mx.crossproduct.snp_crossprod(X, k, rows, is_snpmajor=False, is_plink_format=True, out=M)
    dt = time.perf_counter() - t0
    ok = M[5, 7] == M[7, 5] and M[rows - 1, rows - 1] > 0
    print(f"snp_multiply_gpu host->host k={k} rows={rows} ({X.nbytes/1e9:.1f} GB in, {M.nbytes/1e9:.1f} GB out) {label}: {dt:.2f} s = {2.0*k*rows*rows/dt*1e-15:.2f} Pop/s PCIe-inclusive{'' if ok else '  RESULT WRONG'}", flush=True)


for r in range(reps):
    run("fresh result, pages populated in the background (default)", {}, True)
    run("fresh result, no prefault (round 3 behaviour)", {"MXA_PREFAULT_THREADS": "0"}, True)
    run("fresh result, prefault without the huge-page hint", {"MXA_HOST_HUGEPAGE": "0"}, True)
    run("result pages already touched", {}, False)
run("unpipelined, pages already touched", {"MXA_XPROD_NO_PIPELINE": "1"}, False)
