#!/usr/bin/env python
"""Timeline of ONE training step from a rocprofv3 rocpd database: every kernel dispatch between two optimizer launches,
in start order, with its queue, start offset, duration and the idle gap before it (per queue).
    python tools/rocpd_timeline.py gpurun_out/prof/x_results.db [step_index_from_end=2] [delimiter substring]"""
import sqlite3
import sys
import collections

db = sqlite3.connect(sys.argv[1])
back = int(sys.argv[2]) if len(sys.argv) > 2 else 2
delim = sys.argv[3] if len(sys.argv) > 3 else "adam"
cols = [r[1] for r in db.execute("pragma table_info(rocpd_kernel_dispatch)")]
qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
rows = db.execute(f"""select s.kernel_name, d.start, d.end, {('d.' + qcol) if qcol else '0'}
                      from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id order by d.start""").fetchall()
marks = [i for i, r in enumerate(rows) if delim in r[0].lower()]
# group consecutive delimiter launches (foreach Adam = several kernels) into one mark
groups = []
for i in marks:
    if groups and i - groups[-1][-1] <= 12:
        groups[-1].append(i)
    else:
        groups.append([i])
lo, hi = groups[-back - 1][-1] + 1, groups[-back][0]
step = rows[lo:hi]
t0 = step[0][1]
print(f"# step of {len(step)} dispatches, {(step[-1][2] - t0) / 1e6:.3f} ms from first start to last end")
last_end = {}
busy = collections.Counter()
bykernel = collections.Counter()
for name, s, e, q in step:
    gap = (s - last_end[q]) / 1e3 if q in last_end else 0.0
    last_end[q] = e
    short = name.replace("zsv::", "").split("(")[0][:70]
    busy[q] += e - s
    bykernel[short] += e - s
    print(f"{(s - t0) / 1e6:9.3f} ms  q{q}  {(e - s) / 1e3:9.1f} us  gap {gap:8.1f} us  {short}")
print("# busy per queue (ms):", {q: round(v / 1e6, 3) for q, v in busy.items()})
print("# by kernel (ms):")
for k, v in bykernel.most_common(40):
    print(f"#   {v / 1e6:8.3f}  {k}")
