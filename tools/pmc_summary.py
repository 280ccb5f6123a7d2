#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection CSVs per kernel:
    python tools/pmc_summary.py gpurun_out/pmc_r01/*/*counter_collection.csv
Prints mean counter value per dispatch for every (kernel, counter)."""
import collections
import csv
import sys

acc = collections.defaultdict(lambda: [0.0, 0])
for path in sys.argv[1:]:
    with open(path) as f:
        for r in csv.DictReader(f):
            k = r.get("Kernel_Name", "")[:60]
            c = r.get("Counter_Name", "")
            v = float(r.get("Counter_Value", 0) or 0)
            a = acc[(k, c)]
            a[0] += v
            a[1] += 1
kernels = sorted({k for k, _ in acc})
for k in kernels:
    items = {c: a for (kk, c), a in acc.items() if kk == k}
    n = max(a[1] for a in items.values())
    if n == 0:
        continue
    print(f"{k}  (dispatches {n})")
    for c, a in sorted(items.items()):
        print(f"    {c:36s} {a[0] / a[1]:18.1f}")
