#!/usr/bin/env python3
"""Per-kernel durations of the LAST UNet forward in a rocprofv3 --kernel-trace database (rocpd sqlite), plus the
idle gaps between consecutive kernels: tools/trace_forward.py results.db [first-kernel-substring]"""
import collections
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
first = sys.argv[2] if len(sys.argv) > 2 else "sinusoid"
rows = list(db.execute("select name, start, end, duration, grid_x, grid_y, grid_z, workgroup_x from kernels order by start"))
idx = [i for i, r in enumerate(rows) if first in r[0]]
starts = [i for k, i in enumerate(idx) if k == 0 or i - idx[k - 1] > 8]        # first launch of each forward
last = rows[starts[-2]:starts[-1]] if len(starts) > 1 else rows[starts[-1]:]
print(len(last), "launches in the forward; span %.3f ms; sum of kernel durations %.3f ms" %
      ((last[-1][2] - last[0][1]) / 1e6, sum(r[3] for r in last) / 1e6))
agg = collections.defaultdict(lambda: [0, 0])
for name, s, e, d, gx, gy, gz, wx in last:
    short = re.sub(r"\(anonymous namespace\)::", "", name)
    short = re.sub(r"^void ", "", short)
    short = re.sub(r"^sd::", "", short)
    short = short.split("(")[0]
    agg[short][0] += 1
    agg[short][1] += d
for k, (n, s) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{k[:78]:78s} {n:4d} {s / 1e3:8.1f} us  avg {s / n / 1e3:7.2f}")
gaps = [last[i + 1][1] - last[i][2] for i in range(len(last) - 1)]
print("sum of gaps %.3f ms, mean %.2f us" % (sum(gaps) / 1e6, sum(gaps) / len(gaps) / 1e3))
