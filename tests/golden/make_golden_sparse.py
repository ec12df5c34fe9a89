#!/usr/bin/env python3
"""Emit tests/golden/sparse_golden.npz from the REFERENCE's own CPU library (oracle/_ref, build container only):
inputs (PLINK bytes of both orientations, zero-based CSR) and what the reference's sparse_times_plink('N', tc, ...) wrote.
Each case runs in a child process (the reference exits on errors).  Fixtures are data only."""
import ctypes
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from _util import REF_SO, have_reference, make_problem, random_csr  # noqa: E402

CASES = [
    # name, snps, indiv, nIdx, max_nnz, ldc_pad, missing_frac
    ("small_40x64", 40, 64, 5, 11, 0, 0.0),
    ("odd_1003x501", 1003, 501, 37, 9, 0, 0.0),
    ("ld_777x1301", 777, 1301, 20, 30, 3, 0.0),
    ("missing_600x203", 600, 203, 16, 40, 0, 0.1),
]


def child(path):
    d = np.load(path)
    L = ctypes.CDLL(REF_SO, mode=os.RTLD_LAZY | os.RTLD_GLOBAL)
    vp = ctypes.c_void_p
    L.setOptions_compressed(0, 4, 0, 0, 1, 1, 0, 0, 256, 0)
    L.sparse_times_plink.argtypes = [ctypes.c_char_p, ctypes.c_char_p, vp, vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, vp, vp, vp, vp, ctypes.c_int]
    snps, indiv, ldc = int(d["dims"][0]), int(d["dims"][1]), int(d["dims"][2])
    out = {}
    for tc in ("N", "T"):
        ia, ja, a = d[f"ia{tc}"], d[f"ja{tc}"], d[f"a{tc}"]
        entries = snps if tc == "T" else indiv
        C = np.full((entries, ldc), -777.0)
        plink, plink_t = np.ascontiguousarray(d["plink"]), np.ascontiguousarray(d["plink_t"])
        L.sparse_times_plink(b"N", tc.encode(), plink.ctypes.data_as(vp), plink_t.ctypes.data_as(vp), snps, indiv, len(ia) - 1,
                             ia.ctypes.data_as(vp), ja.ctypes.data_as(vp), a.ctypes.data_as(vp), C.ctypes.data_as(vp), ldc)
        out[f"C{tc}"] = C
    np.savez(path + ".out.npz", **out)


def main():
    assert have_reference(), "build the reference first: make -C oracle ref"
    out, names = {}, []
    for name, snps, indiv, nidx, max_nnz, ldc_pad, miss in CASES:
        prob = make_problem(snps, indiv, 1, seed=sum(map(ord, name)), missing_frac=miss)
        case = {"plink": prob["plink"], "plink_t": prob["plink_t"], "dims": np.array([snps, indiv, nidx + ldc_pad, nidx], np.int64)}
        for tc in ("N", "T"):
            rows = indiv if tc == "T" else snps
            ia, ja, a = random_csr(nidx, rows, max_nnz, seed=len(name) + (tc == "T"))
            case[f"ia{tc}"], case[f"ja{tc}"], case[f"a{tc}"] = ia, ja, a
        tmp = os.path.join("/tmp", f"sparse_case_{name}.npz")
        np.savez(tmp, **case)
        subprocess.run([sys.executable, os.path.abspath(__file__), "--child", tmp], check=True, stdout=subprocess.DEVNULL, timeout=300)
        res = np.load(tmp + ".out.npz")
        for k, v in case.items():
            out[f"{name}/{k}"] = v
        for tc in ("N", "T"):
            out[f"{name}/C{tc}"] = res[f"C{tc}"]
        names.append(name)
        os.remove(tmp); os.remove(tmp + ".out.npz")
        print("done", name)
    out["names"] = np.array(names)
    dst = os.path.join(HERE, "sparse_golden.npz")
    np.savez_compressed(dst, **out)
    print("wrote", dst, os.path.getsize(dst), "bytes")


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--child":
        child(sys.argv[2])
    else:
        main()
