#!/usr/bin/env python3
"""Per-kernel averages of rocprofv3 --pmc CSV output (counter_collection.csv).
   python tools/pmc_summary.py <dir> [kernel-substring]"""
import csv, glob, os, sys
from collections import defaultdict
d = sys.argv[1]; filt = sys.argv[2] if len(sys.argv) > 2 else "wm::"
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if filt not in k:
            continue
        acc[k.split("(")[0][:70]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, cs in acc.items():
    print(k)
    for c, v in sorted(cs.items()):
        print(f"    {c:28s} n={len(v):3d}  mean {sum(v) / len(v):18.1f}")
