#!/usr/bin/env python3
"""Sum rocprofv3 --pmc counter_collection CSVs per kernel (development aid).
usage: pmc_summary.py p_counter_collection.csv [more.csv ...]"""
import csv
import re
import sys
from collections import defaultdict

agg = defaultdict(lambda: defaultdict(float))
calls = defaultdict(set)
for path in sys.argv[1:]:
    for r in csv.DictReader(open(path)):
        m = re.search(r"(k_[a-z_0-9]+)(<[^>]*>)?", r["Kernel_Name"])
        k = (m.group(1) + (m.group(2) or "")) if m else r["Kernel_Name"][:40]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        calls[k].add(r["Dispatch_Id"])
names = sorted({c for k in agg for c in agg[k]})
print("kernel".ljust(30), "calls", *[n.replace("SQ_", "")[:16].rjust(17) for n in names])
for k in sorted(agg, key=lambda k: -agg[k].get("SQ_WAVE_CYCLES", agg[k].get(names[0], 0))):
    print(k.ljust(30), str(len(calls[k])).rjust(5), *[f"{agg[k].get(n, 0):17.4g}" for n in names])
