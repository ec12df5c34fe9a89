#!/usr/bin/env python3
"""repeatability of the host-result paths: snp_multiply_gpu (slab pipeline), mxa_grm fused (slab pipeline) and unfused (one copy at the end).
usage: perf_host_repeat.py snps indiv reps"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import miraculix_amd as mx
from bench import synth_plink_device
snps, indiv, reps = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
L = mx.load_shared_library(); P = mx.lib.ptr
Xh = synth_plink_device(torch, indiv, (snps + 3) // 4, 7, torch.device("cuda", 0)).cpu().numpy()
fh = np.random.default_rng(0).uniform(0.1, 0.5, snps)
Gh = np.zeros((indiv, indiv)); Gh.fill(0.0)
def t(fn):
    t0 = time.perf_counter(); assert fn() == 0; return (time.perf_counter() - t0) * 1e3
for label, env, fn in (("snp_multiply_gpu pipelined", {}, lambda: L.snp_multiply_gpu(P(Xh), snps, indiv, P(Gh), True)),
                       ("snp_multiply_gpu one copy", {"MXA_XPROD_NO_PIPELINE": "1"}, lambda: L.snp_multiply_gpu(P(Xh), snps, indiv, P(Gh), True)),
                       ("mxa_grm fused pipelined", {}, lambda: L.mxa_grm(P(Xh), snps, indiv, P(Gh), 1, 1, P(fh))),
                       ("mxa_grm fused one copy", {"MXA_XPROD_NO_PIPELINE": "1"}, lambda: L.mxa_grm(P(Xh), snps, indiv, P(Gh), 1, 1, P(fh))),
                       ("mxa_grm unfused one copy", {"MXA_XPROD_FUSED_POST": "0"}, lambda: L.mxa_grm(P(Xh), snps, indiv, P(Gh), 1, 1, P(fh)))):
    for k in ("MXA_XPROD_NO_PIPELINE", "MXA_XPROD_FUSED_POST"): os.environ.pop(k, None)
    os.environ.update(env)
    print(f"{label} ({Gh.nbytes/1e9:.1f} GB out): " + " ".join(f"{t(fn):.0f}" for _ in range(reps)) + " ms", flush=True)
