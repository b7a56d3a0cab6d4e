#!/usr/bin/env python
"""Per-kernel time inside each (shape, kind) group of a `rocprofv3 --kernel-trace -- python3 tools/conv_bench.py --markers ...`
run: the trace (rocpd database) is split at the one-element fill launches conv_bench puts in front of every group, the labels come
from the MARK lines of its stdout.
    python tools/kernel_breakdown.py <x_results.db> <conv_bench stdout> [warm-up calls per group = 2]"""
import collections
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
labels = [l.split()[1:] for l in open(sys.argv[2]) if l.startswith("MARK ")]
rows = db.execute("""select s.kernel_name, d.start, d.end from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s
                     on d.kernel_id = s.id order by d.start""").fetchall()
groups, cur = [], None
for name, s, e in rows:
    if "FillFunctor" in name:
        cur = []
        groups.append(cur)
    elif cur is not None and "zsv" in name:
        cur.append((name.replace("zsv::", "").split("(")[0].replace("void ", "")[:64], e - s))
if len(groups) != len(labels):
    print(f"# {len(groups)} marker groups in the trace, {len(labels)} MARK lines", file=sys.stderr)
for (shape, kind, flops), g in zip(labels, groups):
    by = collections.OrderedDict()
    for name, d in g:
        by.setdefault(name, []).append(d)
    # each call launches the same kernel sequence: per-call time = group total / calls, calls = launches of the rarest kernel
    calls = min(len(v) for v in by.values()) if by else 1
    total = sum(sum(v) for v in by.values()) / calls / 1e3
    print(f"{shape:4s} {kind:6s} sum of kernels {total:8.1f} us/call  ({float(flops) / total / 1e6:6.1f} TFLOP/s alg)  calls {calls}")
    for name, v in by.items():
        print(f"      {sum(v) / calls / 1e3:8.1f} us  x{len(v) / calls:4.1f}  {name}")
