#!/usr/bin/env python
"""HBM bytes per kernel over a profiled run: sums FETCH_SIZE (x2, KiB) and WRITE_SIZE (KiB) of two `rocprofv3 --kernel-trace --pmc X
--output-format csv` passes by kernel name:  python tools/pmc_step_traffic.py <fetch dir> <write dir> [passes (fwd+bwd passes in the run)]"""
import csv
import glob
import os
import sys
import collections


def load(root, counter):
    tot, calls = collections.Counter(), collections.Counter()
    for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                name = r["Kernel_Name"].replace("zsv::", "").split("(")[0].replace("void ", "")
                tot[name] += float(r["Counter_Value"])
                calls[name] += 1
    return tot, calls


fetch, calls = load(sys.argv[1], "FETCH_SIZE")
write, _ = load(sys.argv[2], "WRITE_SIZE")
passes = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
rows = sorted(((fetch[k] * 2 + write.get(k, 0.0)) * 1024, k) for k in fetch)
total = sum(b for b, _ in rows)
print(f"# total {total / passes / 1e9:.2f} GB per pass ({passes:g} passes)")
for b, k in reversed(rows[-40:]):
    print(f"{b / passes / 1e6:10.1f} MB  read {fetch[k] * 2048 / passes / 1e6:9.1f}  write {write.get(k, 0.0) * 1024 / passes / 1e6:9.1f}  calls {calls[k] / passes:6.1f}  {k[:90]}")
