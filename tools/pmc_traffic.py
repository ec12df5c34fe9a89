#!/usr/bin/env python3
"""Two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs, --kernel-trace only) -> fabric-side bytes per launch of the kernels whose name
contains <pattern>.  usage: pmc_traffic.py <fetch_dir> <write_dir> <pattern> [out.json]
Units / corrections per /opt/skills/guides/MI355X_MICROARCH.md (HBM section): the counters are KiB; on gfx950 FETCH_SIZE reports half the bytes of a wide
coalesced stream, so the read side is multiplied by the factor calibrated on a known-size stream through this library's LDS-DMA units
(profiles/r02_pmc_calibration.json: 2.000; default 2 when the file is absent); WRITE_SIZE is exact for 16-byte-per-lane stores."""
import collections, csv, glob, json, os, sys

fetch_dir, write_dir, pat = sys.argv[1], sys.argv[2], sys.argv[3]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
factor = 2.0
try:
    factor = float(json.load(open(os.path.join(ROOT, "profiles", "r02_pmc_calibration.json")))["fetch_size_factor_lds_dma_1KiB_units"])
except Exception:
    pass


def per_dispatch(d):
    src = glob.glob(d + "/**/*_counter_collection.csv", recursive=True)[0]
    agg = collections.OrderedDict()
    for r in csv.DictReader(open(src)):
        if pat in r["Kernel_Name"]:
            key = (r["Dispatch_Id"], r["Kernel_Name"].split("(")[0][:70])
            agg[key] = agg.get(key, 0.0) + float(r["Counter_Value"])
    return agg


f, w = per_dispatch(fetch_dir), per_dispatch(write_dir)
names = collections.OrderedDict()
for (disp, name), v in f.items():
    names.setdefault(name, {"reads": [], "writes": []})["reads"].append(factor * v * 1024)
for (disp, name), v in w.items():
    names.setdefault(name, {"reads": [], "writes": []})["writes"].append(v * 1024)
out = {"read_factor": factor, "kernels": {}}
for name, d in names.items():
    n = min(len(d["reads"]), len(d["writes"])) or 1
    out["kernels"][name] = {"launches": n, "read_GB_per_launch": round(sum(d["reads"][:n]) / n / 1e9, 4), "write_GB_per_launch": round(sum(d["writes"][:n]) / n / 1e9, 4)}
    print(f"{name:72s} launches {n:3d}  read {out['kernels'][name]['read_GB_per_launch']:10.4f} GB  write {out['kernels'][name]['write_GB_per_launch']:10.4f} GB per launch")
if len(sys.argv) > 4:
    json.dump(out, open(sys.argv[4], "w"), indent=1)
