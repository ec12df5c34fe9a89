#!/usr/bin/env python3
"""Parse the two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs, --kernel-trace only) of bench.py and
write the per-launch HBM traffic of k_gemm to profiles/<tag>_pmc_traffic.json.
Units/corrections per /opt/skills/guides/MI355X_MICROARCH.md §HBM: counters are in KiB; on gfx950 FETCH_SIZE reports
half the bytes of a wide coalesced stream, so the read side is doubled; WRITE_SIZE is exact for 16-B/lane stores."""
import collections, csv, glob, json, sys

tag, fetch_dir, write_dir = sys.argv[1], sys.argv[2], sys.argv[3]

def per_dispatch(d):
    src = glob.glob(d + "/**/*_counter_collection.csv", recursive=True)[0]
    agg = collections.OrderedDict()
    for r in csv.DictReader(open(src)):
        if "k_gemm<" in r["Kernel_Name"]:   # the fp64 kernel, not k_gemm_i8
            agg[r["Dispatch_Id"]] = agg.get(r["Dispatch_Id"], 0.0) + float(r["Counter_Value"])
    return list(agg.values())

f, w = per_dispatch(fetch_dir), per_dispatch(write_dir)
n = min(len(f), len(w))
reads = [2.0 * x * 1024 for x in f[:n]]       # gfx950 correction x2
writes = [x * 1024 for x in w[:n]]
out = {
    "kernel": "k_gemm<8,8>", "launches": n,
    "read_bytes_per_launch_corrected": reads, "write_bytes_per_launch": writes,
    "traffic_bytes_per_launch_avg": (sum(reads) + sum(writes)) / n,
    "note": "launch order alternates 'N','T'; FETCH_SIZE doubled per the gfx950 correction (uncalibrated for the 32-B-per-row LDS-DMA pieces: an upper bound if the correction does not fully apply)",
}
json.dump(out, open(f"profiles/{tag}_pmc_traffic.json", "w"), indent=1)
print(json.dumps(out))
