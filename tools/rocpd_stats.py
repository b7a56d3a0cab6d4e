#!/usr/bin/env python
"""Per-kernel statistics from a rocprofv3 rocpd database (what `--stats` prints as CSV):
    python tools/rocpd_stats.py gpurun_out/prof/bench_results.db [out.csv]"""
import csv
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute("""select s.kernel_name, count(*), sum(d.end - d.start), avg(d.end - d.start), min(d.end - d.start), max(d.end - d.start)
                     from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id
                     group by s.kernel_name order by 3 desc""").fetchall()
total = sum(r[2] for r in rows)
out = [("Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs")]
for name, calls, tot, avg, mn, mx in rows:
    out.append((name, calls, int(tot), round(avg, 1), round(100.0 * tot / total, 3), int(mn), int(mx)))
if len(sys.argv) > 2:
    with open(sys.argv[2], "w", newline="") as f:
        csv.writer(f).writerows(out)
for r in out[:45]:
    print(f"{str(r[0])[:100]:100s} {r[1]:>7} {float(r[2]) / 1e6 if r[1] != 'Calls' else 0:9.2f}ms {r[3]:>10} {r[4]:>7}")
