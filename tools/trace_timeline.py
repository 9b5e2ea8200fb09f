"""Timeline of the LAST forward in a rocprofv3 kernel-trace CSV made from tools/one_forward.py: kernels in start order
with duration and the idle gap before each, plus totals.  usage: trace_timeline.py <kernel_trace.csv> [n_forwards]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
nf = int(sys.argv[2]) if len(sys.argv) > 2 else 4
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# forwards are separated by synchronize(): split at the nf-1 largest gaps among the tail
ks = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows]
# find the last forward: it ends at the last kernel; walk back to the last dec3_synth before it
ends = [i for i, k in enumerate(ks) if "dec3_synth" in k[2]]
last = ks[ends[-2] + 1: ends[-1] + 1]
busy = 0; gap = 0; prev = last[0][0]
for s, e, n in last:
    g = max(0, s - prev)
    print("%8.1f us  gap %6.1f  %s" % ((e - s) / 1e3, g / 1e3, n[:70]))
    busy += e - s; gap += g; prev = max(prev, e)
agg = {}
for s_, e_, n_ in last:
    k = n_.split("(")[0][-48:]
    agg.setdefault(k, [0, 0.0]); agg[k][0] += 1; agg[k][1] += (e_ - s_) / 1e3
print("---- by kernel (last forward) ----")
for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("%8.1f us  x%-3d %s" % (t, c, k))
print("kernels %d  busy %.1f us  gaps %.1f us  span %.1f us" % (len(last), busy / 1e3, gap / 1e3, (last[-1][1] - last[0][0]) / 1e3))
