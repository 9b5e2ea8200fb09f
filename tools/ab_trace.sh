#!/bin/bash
# Same-box per-kernel comparison of two trees (inside gpurun): bash tools/ab_trace.sh <other tree> <tag>
# (B_ENV="NAME=value": an environment setting for the second tree only; A_LIB=<variant library>: tree a runs on it.)
# Kernel trace of 6 single-stream 4K forwards of each tree, alternating twice; prints per-kernel mean us of the last 4 forwards side by side.
set -e
other=$1; tag=${2:-ab}
root=$GRAFT_REPO_ROOT
mkdir -p $root/gpurun_out/$tag
cd /tmp && export TMPDIR=/tmp
for rep in 1 2; do
  for t in a b; do
    if [ $t = a ]; then script=$root/$other/tools/one_forward.py; else script=$root/tools/one_forward.py; fi
    if [ $t = b ] && [ -n "$B_ENV" ]; then export $B_ENV; fi
    if [ $t = a ] && [ -n "$A_LIB" ]; then export LIB=$root/$A_LIB FLDR_LIB=$root/$A_LIB; else unset LIB FLDR_LIB; fi
    NF=6 timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $root/gpurun_out/$tag/$t$rep -- python3 $script > $root/gpurun_out/$tag/$t$rep.log 2>&1
  done
done
python3 - $root/gpurun_out/$tag <<'PY'
import csv, glob, sys, collections, re
root = sys.argv[1]
def load(d):
    f = glob.glob(d + "/*/*_kernel_trace.csv")[-1]
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    # forwards are delimited by the final synthesis kernel
    ends = [i for i, r in enumerate(rows) if "dec23_synth" in r["Kernel_Name"] or "synth_tail" in r["Kernel_Name"]]
    per = collections.defaultdict(float); n = 0
    for k in range(len(ends) - 4, len(ends)):
        for r in rows[ends[k - 1] + 1: ends[k] + 1]:
            name = re.sub(r"\(.*", "", r["Kernel_Name"])[:60]
            per[name] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000.0
        n += 1
    return {k: v / n for k, v in per.items()}
A = [load(root + "/a%d" % r) for r in (1, 2)]; B = [load(root + "/b%d" % r) for r in (1, 2)]
names = sorted(set().union(*A, *B), key=lambda k: -(B[0].get(k, 0)))
ta = [sum(a.values()) for a in A]; tb = [sum(b.values()) for b in B]
print("%-60s %8s %8s %8s %8s %8s" % ("kernel (us per forward, summed over launches)", "a1", "a2", "b1", "b2", "b-a"))
for k in names:
    a = [x.get(k, 0) for x in A]; b = [x.get(k, 0) for x in B]
    print("%-60s %8.1f %8.1f %8.1f %8.1f %+8.1f" % (k, a[0], a[1], b[0], b[1], sum(b) / 2 - sum(a) / 2))
print("%-60s %8.1f %8.1f %8.1f %8.1f %+8.1f" % ("TOTAL", ta[0], ta[1], tb[0], tb[1], sum(tb) / 2 - sum(ta) / 2))
PY
