#!/usr/bin/env python3
"""mxa_grm / mxa_ld with the post-processing fused into the crossproduct epilogue against the three extra passes over the result
(MXA_XPROD_FUSED_POST=0): device-resident result, and host result (fused: through the slab pipeline with its copier threads).
usage: perf_grm.py snps indiv [indiv_for_the_host_case]"""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import miraculix_amd as mx
from bench import synth_plink_device

snps, indiv = int(sys.argv[1]), int(sys.argv[2])
indiv_h = int(sys.argv[3]) if len(sys.argv) > 3 else 0
dev = torch.device("cuda", 0)
L = mx.load_shared_library()
P = mx.lib.ptr


def run(fn, label, reps=2):
    for fused in ("1", "0", "1"):
        os.environ["MXA_XPROD_FUSED_POST"] = fused
        ts = []
        for _ in range(reps):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            assert fn() == 0
            torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        print(f"{label} {'fused epilogue' if fused == '1' else 'three extra passes'}: {min(ts)*1e3:.1f} ms (best of {reps})", flush=True)


X = synth_plink_device(torch, indiv, (snps + 3) // 4, 7, dev)          # individual-major: indiv rows x snps / 4 bytes
f = torch.rand(snps, dtype=torch.float64, device=dev) * 0.4 + 0.1
G = torch.empty((indiv, indiv), dtype=torch.float64, device=dev)
run(lambda: L.mxa_grm(P(X), snps, indiv, P(G), 1, 1, P(f)), f"mxa_grm {snps} SNPs x {indiv} indiv, device result ({G.numel()*8/1e9:.1f} GB)")
keep = G[:3, :3].cpu()
os.environ["MXA_XPROD_FUSED_POST"] = "0"
assert L.mxa_grm(P(X), snps, indiv, P(G), 1, 1, P(f)) == 0
assert torch.equal(keep, G[:3, :3].cpu())
del G
# LD on a SNP-major matrix of the same bytes: `indiv` rows play the SNPs
f2 = torch.rand(indiv, dtype=torch.float64, device=dev) * 0.4 + 0.1
R = torch.empty((indiv, indiv), dtype=torch.float64, device=dev)
run(lambda: L.mxa_ld(P(X), indiv, snps, P(R), 1, P(f2)), f"mxa_ld {indiv} SNPs x {snps} indiv, device result")
del R, X
torch.cuda.empty_cache()
if indiv_h:
    Xh = synth_plink_device(torch, indiv_h, (snps + 3) // 4, 7, dev).cpu().numpy()
    fh = np.random.default_rng(0).uniform(0.1, 0.5, snps)
    Gh = np.zeros((indiv_h, indiv_h)); Gh.fill(0.0)                     # pages touched before the timed calls
    run(lambda: L.mxa_grm(P(Xh), snps, indiv_h, P(Gh), 1, 1, P(fh)), f"mxa_grm {snps} x {indiv_h}, HOST in ({Xh.nbytes/1e9:.1f} GB) / HOST result ({Gh.nbytes/1e9:.1f} GB)", reps=1)
