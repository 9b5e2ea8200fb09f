#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace --stats output directory: per-kernel totals divided by the number of forwards."""
import csv, glob, sys
d, nfwd = sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
f = glob.glob(d + '/*/*kernel_stats.csv')[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print("GPU time per forward: %.3f ms" % (tot / 1e6 / nfwd))
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 24]:
    print("%-66s %6.1f launches/fwd  %7.3f ms/fwd  avg %8.1f us  %5.1f%%" % (r['Name'][:66], float(r['Calls']) / nfwd, float(r['TotalDurationNs']) / 1e6 / nfwd, float(r['AverageNs']) / 1e3, float(r['Percentage'])))
