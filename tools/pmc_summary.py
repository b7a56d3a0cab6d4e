#!/usr/bin/env python
"""Per-kernel means of a `rocprofv3 --kernel-trace --pmc ... --output-format csv` pass (own pass, no other trace domains):
    python tools/pmc_summary.py <dir with *counter_collection.csv> [substring filter]
prints JSON: kernel -> {dispatches, duration_us (mean of the last 3), <counter>: mean of the last 3, mfma_pipe_utilisation, clock_GHz}."""
import csv
import glob
import json
import os
import sys
import collections

root = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else "zsv"
files = glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True)
per = collections.defaultdict(lambda: collections.defaultdict(dict))       # kernel -> dispatch id -> counter -> value
dur = collections.defaultdict(dict)
for f in files:
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        if flt not in name:
            continue
        d = int(r["Dispatch_Id"])
        per[name][d][r["Counter_Name"]] = per[name][d].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        if "Start_Timestamp" in r and r["Start_Timestamp"]:
            dur[name][d] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
out = {}
for name, disp in per.items():
    ids = sorted(disp)[-3:]
    e = {"dispatches": len(disp)}
    if dur[name]:
        e["duration_us"] = round(sum(dur[name][i] for i in ids) / len(ids), 1)
    for c in sorted({c for i in ids for c in disp[i]}):
        e[c] = round(sum(disp[i].get(c, 0.0) for i in ids) / len(ids))
    if "SQ_VALU_MFMA_BUSY_CYCLES" in e and "GRBM_GUI_ACTIVE" in e and e["GRBM_GUI_ACTIVE"]:
        e["mfma_pipe_utilisation"] = round(e["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * e["GRBM_GUI_ACTIVE"] / 8.0), 3)
        if "duration_us" in e:
            e["clock_GHz"] = round(e["GRBM_GUI_ACTIVE"] / 8.0 / e["duration_us"] / 1e3, 2)
    short = name.replace("zsv::", "").split("(")[0].replace("void ", "")
    out[short] = e
print(json.dumps(out, indent=1))
