#!/usr/bin/env python3
"""Pivot rocprofv3 ``*_counter_collection.csv`` files: mean counter value per kernel
(dispatches whose kernel name contains one of the given substrings)."""
import csv
import sys
from collections import defaultdict

files = [a for a in sys.argv[1:] if a.endswith(".csv")]
keys = [a for a in sys.argv[1:] if not a.endswith(".csv")] or ["flash_prefill", "paged_decode"]
acc = defaultdict(lambda: defaultdict(list))
for f in files:
    with open(f) as fh:
        for r in csv.DictReader(fh):
            for k in keys:
                if k in r["Kernel_Name"]:
                    acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in keys:
    print(f"== {k}")
    for c, v in sorted(acc[k].items()):
        v = v[1:] if len(v) > 1 else v          # drop the first (cold) dispatch
        print(f"  {c:34s} {sum(v)/len(v):18.0f}   (n={len(v)})")
