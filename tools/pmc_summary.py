#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection CSVs: mean counter value per kernel name.
Usage: python tools/pmc_summary.py <dir-or-csv> [...]"""
import csv, glob, os, sys, collections
for arg in sys.argv[1:]:
    files = [arg] if arg.endswith(".csv") else glob.glob(os.path.join(arg, "**", "*counter_collection.csv"), recursive=True)
    for f in files:
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            acc[(r["Kernel_Name"].split("(")[0][:70], r["Counter_Name"])].append(float(r["Counter_Value"]))
        print("#", f)
        for (k, c), v in sorted(acc.items()):
            print(f"{k:72s} {c:12s} n={len(v):3d} mean={sum(v)/len(v):16.1f} min={min(v):14.1f} max={max(v):14.1f}")
