#!/usr/bin/env python3
"""Parse the rocprofv3 --pmc passes of tools/pmc_calibrate (known-size streams) and write profiles/r02_pmc_calibration.json.
usage: pmc_calibrate.py <bytes> <fetch_dir> [<rdreq_dir>]"""
import collections, csv, glob, json, sys

nbytes = int(sys.argv[1])
out = {"stream_bytes": nbytes, "method": "every kernel of tools/pmc_calibrate.hip reads the buffer exactly once (8 GiB, 30x the Infinity Cache); "
       "factor = bytes / (counter x unit), per kernel, averaged over its dispatches"}


def per_kernel(d):
    res = collections.defaultdict(lambda: collections.defaultdict(lambda: collections.defaultdict(float)))
    for src in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(src)):
            name = r["Kernel_Name"].split("(")[0]
            res[name][r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
    return res


for d in sys.argv[2:]:
    for kname, ctrs in per_kernel(d).items():
        if "k_stream" not in kname:
            continue
        for cname, disp in ctrs.items():
            vals = list(disp.values())
            avg = sum(vals) / len(vals)
            out.setdefault(kname, {})[cname] = {"per_dispatch_avg": avg, "dispatches": len(vals)}
            if cname == "FETCH_SIZE":
                out[kname][cname]["bytes_over_counter_KiB"] = nbytes / (avg * 1024)
for k, v in list(out.items()):
    if isinstance(v, dict) and "FETCH_SIZE" in v:
        if "k_stream_dma<0>" in k:
            out["fetch_size_factor_lds_dma_1KiB_units"] = round(v["FETCH_SIZE"]["bytes_over_counter_KiB"], 4)
        if "k_stream_dma<1>" in k:
            out["fetch_size_factor_lds_dma_stride32"] = round(v["FETCH_SIZE"]["bytes_over_counter_KiB"], 4)
        if "k_stream_vec" in k:
            out["fetch_size_factor_plain_16B_per_lane"] = round(v["FETCH_SIZE"]["bytes_over_counter_KiB"], 4)
# exact byte count from the request-size split, if that pass was collected: 32 * RDREQ_32B + 64 * RDREQ_64B + 128 * RDREQ_128B
for k, v in list(out.items()):
    if isinstance(v, dict) and "TCC_EA0_RDREQ_128B_sum" in v:
        b = 32 * v.get("TCC_EA0_RDREQ_32B_sum", {"per_dispatch_avg": 0})["per_dispatch_avg"] + 64 * v.get("TCC_EA0_RDREQ_64B_sum", {"per_dispatch_avg": 0})["per_dispatch_avg"] \
            + 128 * v["TCC_EA0_RDREQ_128B_sum"]["per_dispatch_avg"]
        v["bytes_from_request_size_split"] = b
        v["bytes_from_request_size_split_over_true"] = b / nbytes
json.dump(out, open(__import__("os").environ.get("CALIB_OUT", "profiles/r02_pmc_calibration.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
