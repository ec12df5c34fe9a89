#!/usr/bin/env python3
"""random shapes / formats / engines / result residence for the integer crossproduct and the fused GRM / LD against numpy:
snp_multiply_gpu bit-exact against the int64 product (incl. the reference's "byte with a missing pair -> 0xFF" table, snp_multiply_cuda.h:202-210),
mxa_grm / mxa_ld fused bit-identical to the unfused passes and within 1e-12 of the dense restatement of crossproduct.jl:83-152.
fuzz_crossprod.py [cases] [seed]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import miraculix_amd as mx

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 120
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
mx.load_shared_library()
cp = mx.crossproduct
dev = torch.device("cuda", 0)
PLINK_CODE = np.array([0, 2, 3, 1], dtype=np.uint8)     # value 0 / 1 / 2 / missing -> PLINK bits 00 / 10 / 11 / 01


def pack(fields):
    rows, k = fields.shape
    P = np.zeros((rows, (k + 3) // 4 * 4), dtype=np.uint8); P[:, :k] = fields
    return np.ascontiguousarray(P[:, 0::4] | (P[:, 1::4] << 2) | (P[:, 2::4] << 4) | (P[:, 3::4] << 6))


worst = 0.0
for c in range(cases):
    rows = int(rng.choice([rng.integers(1, 40), rng.integers(40, 600), rng.integers(600, 1800)]))
    k = int(rng.choice([rng.integers(1, 130), rng.integers(130, 900), rng.integers(900, 6000)]))
    plink = bool(rng.integers(0, 2))
    eng = str(rng.choice(["f4", "i8"]))
    os.environ["MXA_XPROD_ENGINE"] = eng
    if plink:
        Z = rng.integers(0, 3, size=(rows, k)).astype(np.int64)
        miss = rng.random((rows, k)) < float(rng.choice([0.0, 0.0, 0.01]))
        codes = PLINK_CODE[np.where(miss, 3, Z)]
        X = pack(codes)
        # the reference's byte table: a byte holding a missing pair becomes 0xFF, i.e. all four of its fields count 3 (also the padding fields of the last byte)
        kp = X.shape[1] * 4
        Zp = np.zeros((rows, kp), dtype=np.int64); Zp[:, :k] = Z
        mp = np.zeros((rows, kp), dtype=bool); mp[:, :k] = miss
        bad = mp.reshape(rows, -1, 4).any(axis=2).repeat(4, axis=1)
        V = np.where(bad, 3, Zp)
    else:
        V = rng.integers(0, 4, size=(rows, k)).astype(np.int64)
        X = pack(V.astype(np.uint8))
    ref = (V @ V.T).astype(np.float64)
    on_dev = bool(rng.integers(0, 2))
    Xa = torch.from_numpy(X).to(dev) if on_dev else X
    if rng.integers(0, 3) == 0 and not on_dev:
        os.environ["MXA_XPROD_SLAB_MB"] = "1"
    else:
        os.environ.pop("MXA_XPROD_SLAB_MB", None)
    M = cp.snp_crossprod(Xa, k, rows, is_snpmajor=False, is_plink_format=plink)
    M = M.cpu().numpy() if on_dev else M
    if not np.array_equal(M, ref):
        print(f"FAIL crossprod case {c}: rows={rows} k={k} plink={plink} engine={eng} device={on_dev} max diff {np.abs(M - ref).max()}", flush=True); sys.exit(1)
    # GRM (rows = individuals) and LD (rows = SNPs) on the same packed matrix
    f = rng.uniform(0.05, 0.5, size=k)
    Vf = V[:, :k].astype(np.float64) if V.shape[1] != k else V.astype(np.float64)
    if plink and V.shape[1] != k:      # padding fields of a 0xFF byte belong to the product (they are real fields of the staged matrix)
        Vf = V.astype(np.float64)
    res = {}
    for fused in ("1", "0"):
        os.environ["MXA_XPROD_FUSED_POST"] = fused
        G = cp.grm(Xa, k, rows, is_plink_format=plink, do_scale=True, allele_freq=torch.from_numpy(f).to(dev) if on_dev else f)
        res[fused] = G.cpu().numpy() if on_dev else G
    os.environ.pop("MXA_XPROD_FUSED_POST")
    if not np.array_equal(res["1"], res["0"], equal_nan=True):
        print(f"FAIL grm fused != unfused case {c}: rows={rows} k={k} plink={plink} engine={eng}", flush=True); sys.exit(1)
    Mx = Vf @ Vf.T
    cs = Mx.sum(axis=0)
    Gref = (Mx - np.outer(cs, np.ones(rows)) / rows - np.outer(np.ones(rows), cs) / rows + cs.sum() / rows ** 2) / (2 * np.sum(f * (1 - f)))
    err = np.abs(res["1"] - Gref).max() / max(np.abs(Gref).max(), 1e-300)
    worst = max(worst, err)
    if not err <= 1e-12:
        print(f"FAIL grm vs dense case {c}: rows={rows} k={k} plink={plink} engine={eng} err={err:.3e}", flush=True); sys.exit(1)
    if (c + 1) % 20 == 0:
        print(f"fuzz_crossprod: {c + 1} / {cases} cases, worst GRM error {worst:.2e}", flush=True)
print(f"fuzz_crossprod: {cases} cases ok (crossproduct bit-exact, GRM fused == unfused, worst GRM error vs dense {worst:.2e})")
