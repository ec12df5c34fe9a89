"""per-call wall time of dgemm_compressed with HOST B / C (the plain ABI), looking for outliers: python tools/perf_host_call_outliers.py snps indiv n reps"""
import sys, time
import numpy as np
sys.path.insert(0, ".")
import miraculix_amd as mx
import torch

snps, indiv, n, reps = (int(x) for x in sys.argv[1:5])
mx.load_shared_library()
dg = mx.dgemm_compressed
g = torch.Generator(device="cuda"); g.manual_seed(1)
plink = torch.randint(0, 256, (snps, (indiv + 3) // 4), dtype=torch.uint8, device="cuda", generator=g)
f = np.full(snps, 0.3)
dg.set_options(use_gpu=True, not_center=False, verbose=0)
obj = dg.init_compressed(plink, None, snps, indiv, f, n)
rng = np.random.default_rng(0)
for trans in (False, True):
    k = indiv if trans else snps
    B = np.asfortranarray(rng.standard_normal((k, n)))
    ts = []
    for i in range(reps):
        t0 = time.perf_counter()
        C = dg.dgemm_compressed_main(trans, obj, B, snps, indiv)
        ts.append(time.perf_counter() - t0)
    ts = np.array(ts) * 1e3
    med = np.median(ts[1:])
    out = [(i, round(float(t), 2)) for i, t in enumerate(ts) if i > 0 and t > 2 * med]
    print(f"{'T' if trans else 'N'}: median {med:.3f} ms, calls above 2x median (index, ms): {out}", flush=True)
dg.free_compressed(obj)
# control: the bare pageable upload of the same B through torch (no library call): outliers here belong to the runtime's pageable-copy path
B = np.asfortranarray(rng.standard_normal((snps, n)))
dst = torch.empty((n, snps), dtype=torch.float64, device="cuda")
src = torch.from_numpy(B.T)
ts = []
for i in range(reps):
    t0 = time.perf_counter()
    dst.copy_(src); torch.cuda.synchronize()
    ts.append(time.perf_counter() - t0)
ts = np.array(ts) * 1e3
med = np.median(ts[1:])
print(f"bare pageable upload of {B.nbytes / 1e6:.0f} MB: median {med:.3f} ms, copies above 2x median (index, ms): {[(i, round(float(t), 2)) for i, t in enumerate(ts) if i > 0 and t > 2 * med]}", flush=True)
