#!/usr/bin/env python3
"""Median per-launch value of every counter in a rocprofv3 --pmc output directory for kernels whose name contains <substr>.
usage: pmc_kernel.py <dir> <substr>"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*counter_collection.csv")[0]
vals = {}
for r in csv.DictReader(open(f)):
    if sys.argv[2] in r["Kernel_Name"]:
        vals.setdefault(r["Counter_Name"], {}).setdefault(r["Dispatch_Id"], 0.0)
        vals[r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
for c, d in sorted(vals.items()):
    v = sorted(d.values())
    print("%-32s %16.0f   (%d launches)" % (c, v[len(v) // 2], len(v)))
