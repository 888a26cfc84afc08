#!/usr/bin/env python3
"""Per-kernel averages of rocprofv3 --pmc counter_collection.csv files: python tools/pmc_table.py DIR... [--kernel SUBSTR]"""
import collections
import csv
import glob
import sys

args = [a for a in sys.argv[1:] if not a.startswith("--")]
want = next((a.split("=", 1)[1] for a in sys.argv[1:] if a.startswith("--kernel=")), "")
tab = collections.defaultdict(lambda: collections.defaultdict(list))
for d in args:
    for p in glob.glob(d + "/*/*counter_collection.csv") + glob.glob(d + "/*counter_collection.csv"):
        for r in csv.DictReader(open(p)):
            if want in r["Kernel_Name"]:
                tab[r["Kernel_Name"][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in tab.items():
    print(k)
    for c, v in sorted(cs.items()):
        print(f"   {c:32s} n={len(v):3d} avg={sum(v) / len(v):16.1f}")
