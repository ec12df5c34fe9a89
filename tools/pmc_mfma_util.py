#!/usr/bin/env python3
"""rocprofv3 --pmc SQ_* pass -> MFMA-pipe utilisation per launch of the kernels whose name contains <pattern>.
usage: pmc_mfma_util.py <dir with *_counter_collection.csv and *_kernel_trace.csv> <pattern> <out.json> [mfma_cycles_per_instr]
mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (duration x effective clock x 1024 SIMDs); effective clock = GRBM_GUI_ACTIVE / duration."""
import collections, csv, glob, json, sys

d, pat, out = sys.argv[1], sys.argv[2], sys.argv[3]
cc = glob.glob(d + "/**/*_counter_collection.csv", recursive=True)[0]
kt = glob.glob(d + "/**/*_kernel_trace.csv", recursive=True)[0]
dur = {}
for r in csv.DictReader(open(kt)):
    if pat in r["Kernel_Name"]:
        dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6
vals = collections.defaultdict(dict)
for r in csv.DictReader(open(cc)):
    if pat in r["Kernel_Name"]:
        vals[r["Dispatch_Id"]][r["Counter_Name"]] = vals[r["Dispatch_Id"]].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
launches = []
for disp, c in vals.items():
    if disp not in dur:
        continue
    ms = dur[disp]
    # GRBM_GUI_ACTIVE is reported once per XCD; the rows of a dispatch were summed above, so divide by the 8 XCDs of an MI355X
    clk = c.get("GRBM_GUI_ACTIVE", 0.0) / 8.0 / (ms * 1e-3) * 1e-9 if c.get("GRBM_GUI_ACTIVE") else None
    e = {"dispatch": disp, "duration_ms": ms, "eff_clock_GHz": clk, "mfma_busy_cycles": c.get("SQ_VALU_MFMA_BUSY_CYCLES")}
    if clk and c.get("SQ_VALU_MFMA_BUSY_CYCLES"):
        e["mfma_util"] = c["SQ_VALU_MFMA_BUSY_CYCLES"] / (ms * 1e-3 * clk * 1e9 * 1024)
    if c.get("SQ_WAVE_CYCLES"):
        for k, name in (("SQ_WAIT_ANY", "wait_any"), ("SQ_WAIT_INST_ANY", "wait_inst_any"), ("SQ_ACTIVE_INST_ANY", "active_inst_any")):
            if k in c:
                e[name] = c[k] / c["SQ_WAVE_CYCLES"]
    for k, name in (("SQ_INSTS_VALU", "insts_valu"), ("SQ_INSTS_LDS", "insts_lds")):
        if k in c:
            e[name] = c[k]
    launches.append(e)
res = {"kernel_pattern": pat, "note": "rocprofv3 --pmc pass (profiled runs clock a few % lower); mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (duration x effective clock x 1024 SIMDs)", "launches": launches}
json.dump(res, open(out, "w"), indent=1)
u = [e["mfma_util"] for e in launches if "mfma_util" in e]
print(pat, "launches", len(launches), "mfma_util", [round(x, 3) for x in u], "clock", [round(e["eff_clock_GHz"], 3) for e in launches if e["eff_clock_GHz"]])
