#!/usr/bin/env python3
"""rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE output (counter_collection.csv or the rocpd *_results.db; two separate passes) ->
profiles/rNN_pmc_hbm_traffic_per_launch.json, keyed by the kernel names bench.py reports.
FETCH_SIZE is doubled: on gfx950 it reports half of the bytes of wide coalesced reads
(/opt/skills/guides/MI355X_MICROARCH.md, HBM section)."""
import collections
import csv
import glob
import json
import re
import sqlite3
import sys

fetch_dir, write_dir, out = sys.argv[1], sys.argv[2], sys.argv[3]


def agg(d, counter):
    a = collections.defaultdict(lambda: [0.0, 0])
    for path in glob.glob(d + "/*/*counter_collection.csv"):
        for r in csv.DictReader(open(path)):
            if r["Counter_Name"] == counter:
                a[r["Kernel_Name"]][0] += float(r["Counter_Value"])
                a[r["Kernel_Name"]][1] += 1
    for path in glob.glob(d + "/*/*_results.db") + glob.glob(d + "/*_results.db"):
        db = sqlite3.connect(path)
        for name, value in db.execute("select kernel_name, value from counters_collection where counter_name = ?",
                                      (counter,)):
            a[name][0] += float(value)
            a[name][1] += 1
    return a


def short(k):
    m = re.search(r"(igemm2_kernel<[^>]*>|igemm_kernel<[^>]*>|splitk_epilogue_kernel)", k)
    if m:
        return m.group(1).replace(" ", "")
    m = re.search(r"N_1\d\d?([a-z_0-9]+?_kernel)(ILi(\d+))?", k)
    if m:
        return m.group(1) + (f"<{m.group(3)}>" if m.group(3) else "")
    m = re.search(r"(\w+_kernel)\(", k)          # demangled, non-template: sd::(anonymous namespace)::name(args)
    if m:
        return m.group(1)
    return k[:60]


f, w = agg(fetch_dir, "FETCH_SIZE"), agg(write_dir, "WRITE_SIZE")
res = {"_note": "HBM bytes per launch, rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes) over "
                "`bench.py --steps 1 --warmup 0 --denoise-steps 2 --no-cpu-baseline --no-roofline`; kB * 1024, "
                "FETCH_SIZE doubled for gfx950; averages over every launch of the kernel (UNet + VAE shapes mixed)."}
for k in f:
    if "sd" not in k:
        continue
    fs, n = f[k]
    ws, n2 = w.get(k, [0, 1])
    res[short(k)] = {"launches": n, "fetch_MB_per_launch": round(fs / n * 1024 * 2 / 1e6, 3),
                     "write_MB_per_launch": round(ws / max(n2, 1) * 1024 / 1e6, 3)}
json.dump(res, open(out, "w"), indent=1)
print("wrote", out, len(res) - 1, "kernels")
