#!/usr/bin/env python3
"""Object life-cycle soak: create / use / free every kind of object many times (single-device, multi-shard, from a .bed range), run the
crossproduct entries in between, and check that device memory returns to where it started (leaks) and that results stay correct."""
import ctypes, os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import miraculix_amd as mx
from _util import make_problem

L = mx.load_shared_library()
dg = mx.dgemm_compressed
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 40
only = sys.argv[2] if len(sys.argv) > 2 else "all"      # all | kind0 | kind1 | kind2 | kind3 | xprod | nomul
snps, indiv, n = 6001, 1201, 10
prob = make_problem(snps, indiv, n, seed=3)
Z = prob["Z"].astype(np.float64)
rng = np.random.default_rng(0)
tmp = tempfile.mkdtemp()
bed = os.path.join(tmp, "x.bed")
mx.read_plink.write_bed(bed, prob["plink"])
torch.cuda.init()
dg.set_options(use_gpu=True, not_center=True, verbose=0)


def free_bytes():
    torch.cuda.synchronize()
    return torch.cuda.mem_get_info()[0]


base = None
t0 = time.time()
for it in range(iters):
    kind = it % 4 if only in ("all", "xprod", "nomul") else int(only[-1])
    if only == "xprod":
        obj = None
    elif kind == 0:
        obj = dg.init_compressed(prob["plink"], prob["plink_t"], snps, indiv, prob["f"], n)
        s0, s1 = 0, snps
    elif kind == 1:
        os.environ["MIRACULIX_NUM_GPUS"] = str(2 + it % 3)
        obj = dg.init_compressed(prob["plink"], prob["plink_t"], snps, indiv, prob["f"], n)
        os.environ.pop("MIRACULIX_NUM_GPUS")
        s0, s1 = 0, snps
    elif kind == 2:
        s0, s1 = 1000, 5003
        obj, _ = dg.init_compressed_from_bed_range(bed, s0, s1, n, snps=snps, indiv=indiv)
    else:
        os.environ["MIRACULIX_NUM_GPUS"] = "3"
        obj, _, _, _ = dg.init_compressed_from_bed(bed, n, snps=snps, indiv=indiv)
        os.environ.pop("MIRACULIX_NUM_GPUS")
        s0, s1 = 0, snps
    for nn in ((1, 2, n, 33) if obj is not None and only != "nomul" else ()):
        B = rng.standard_normal((s1 - s0, nn))
        C = dg.dgemm_compressed_main(False, obj, np.asfortranarray(B), s1 - s0, indiv)
        ref = Z[:, s0:s1] @ B
        assert np.abs(C - ref).max() <= 1e-11 * np.abs(ref).max(), (it, kind, nn)
        Bt = rng.standard_normal((indiv, nn))
        Ct = dg.dgemm_compressed_main(True, obj, np.asfortranarray(Bt), s1 - s0, indiv)
        reft = Z[:, s0:s1].T @ Bt
        assert np.abs(Ct - reft).max() <= 1e-11 * np.abs(reft).max(), (it, kind, nn)
    if obj is not None:
        dg.free_compressed(obj)
    if only in ("all", "xprod"):
        X = prob["plink_t"][: 300 + it % 50]
        M = mx.crossproduct.snp_crossprod(X, snps, X.shape[0], is_snpmajor=False, is_plink_format=True)
        assert np.array_equal(M, Z[: X.shape[0]] @ Z[: X.shape[0]].T)
    if it == 7:
        base = free_bytes()
end = free_bytes()
print(f"life-cycle soak [{only}]: {iters} objects ok in {time.time() - t0:.1f} s; device memory after object 8: {base / 2**20:.0f} MiB free, at the end: {end / 2**20:.0f} MiB free "
      f"(drift {(base - end) / 2**20:.1f} MiB)")
