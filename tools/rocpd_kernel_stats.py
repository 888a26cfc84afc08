#!/usr/bin/env python3
"""rocprofv3 --kernel-trace --stats writes a rocpd sqlite database on this image; this dumps its
per-kernel summary in the column layout of rocprofv3's kernel_stats.csv so it can be committed under
profiles/.  usage: rocpd_kernel_stats.py <results.db> <out.csv>"""
import csv
import math
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = {}
for name, dur in db.execute("select name, duration from kernels"):
    rows.setdefault(name, []).append(dur)
total = sum(sum(v) for v in rows.values())
with open(sys.argv[2], "w", newline="") as f:
    w = csv.writer(f, quoting=csv.QUOTE_NONNUMERIC)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
    for name, v in sorted(rows.items(), key=lambda kv: -sum(kv[1])):
        n, s = len(v), sum(v)
        mean = s / n
        sd = math.sqrt(sum((x - mean) ** 2 for x in v) / (n - 1)) if n > 1 else 0.0
        w.writerow([name, n, s, round(mean, 6), round(100.0 * s / total, 2), min(v), max(v), round(sd, 6)])
print("wrote", sys.argv[2], len(rows), "kernels")
