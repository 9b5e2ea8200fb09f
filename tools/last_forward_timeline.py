#!/usr/bin/env python3
"""Kernel timeline of the LAST (warm) forward of a `rocprofv3 --kernel-trace` run of tools/one_forward.py:
start offset, duration, kernel name; sum of the kernel times, launch count and span.
usage: last_forward_timeline.py <trace_dir>"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/*/*kernel_trace.csv')[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
start = [i for i, r in enumerate(rows) if 'pcap_init' in r['Kernel_Name']][-1]
t0 = int(rows[start]['Start_Timestamp'])
tot = 0.0
for r in rows[start:]:
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    tot += d
    print("%8.1f %7.1f  %s" % ((int(r['Start_Timestamp']) - t0) / 1e3, d, r['Kernel_Name'][:100]))
print("sum of kernel times %.1f us, %d launches, span %.1f us" % (tot, len(rows) - start, (int(rows[-1]['End_Timestamp']) - t0) / 1e3))
