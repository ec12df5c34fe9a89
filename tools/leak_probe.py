#!/usr/bin/env python3
"""which multi-object operation leaves device memory behind?  usage: leak_probe.py <mode> ; modes: N, T, Ndev, Tdev, gram, n1"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import miraculix_amd as mx
from _util import make_problem
mx.load_shared_library()
dg = mx.dgemm_compressed
mode = sys.argv[1]
shards = int(sys.argv[2]) if len(sys.argv) > 2 else 2
snps, indiv, n = 6001, 1201, 10
prob = make_problem(snps, indiv, n, seed=3)
dg.set_options(use_gpu=True, not_center=True, verbose=0)
rng = np.random.default_rng(0)
BN = np.asfortranarray(rng.standard_normal((snps, n))); BT = np.asfortranarray(rng.standard_normal((indiv, n)))
BNd = torch.from_numpy(np.ascontiguousarray(BN.T)).cuda().t(); BTd = torch.from_numpy(np.ascontiguousarray(BT.T)).cuda().t()
def free(): torch.cuda.synchronize(); return torch.cuda.mem_get_info()[0]
base = None
for it in range(60):
    os.environ["MIRACULIX_NUM_GPUS"] = str(shards)
    if shards == 1: os.environ["MXA_FORCE_MULTI"] = "1"
    obj = dg.init_compressed(prob["plink"], prob["plink_t"], snps, indiv, prob["f"], n)
    os.environ.pop("MIRACULIX_NUM_GPUS"); os.environ.pop("MXA_FORCE_MULTI", None)
    if mode == "N": dg.dgemm_compressed_main(False, obj, BN, snps, indiv)
    if mode == "T": dg.dgemm_compressed_main(True, obj, BT, snps, indiv)
    if mode == "Ndev": dg.dgemm_compressed_main(False, obj, BNd, snps, indiv)
    if mode == "Tdev": dg.dgemm_compressed_main(True, obj, BTd, snps, indiv)
    if mode == "gram": dg.gram_matvec(obj, BT, snps, indiv)
    if mode == "n1": dg.dgemm_compressed_main(True, obj, np.asfortranarray(BT[:, :1]), snps, indiv)
    dg.free_compressed(obj)
    if it == 9: base = free()
print(f"leak probe mode={mode} shards={shards}: drift over 50 objects {(base - free()) / 2**20:.1f} MiB")
