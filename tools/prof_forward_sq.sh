#!/bin/bash
# SQ wave-state counters per kernel of single-stream 4K forwards: where waves wait (memory / issue) and how busy the vector ALUs are.
# usage (inside gpurun): bash tools/prof_forward_sq.sh <tag>   -> gpurun_out/<tag>/sq.txt
tag=${1:-fwd_sq}
export FW=3840 FH=2160 NF=${NF:-3}
root=$GRAFT_REPO_ROOT
out=$root/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAVES SQ_BUSY_CYCLES --output-format csv -d $out/a -- python3 $root/tools/one_forward.py > $out/a.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $out/t -- python3 $root/tools/one_forward.py > $out/t.log 2>&1
python3 - <<PY > $out/sq.txt
import csv, glob, re
from collections import defaultdict
nf = float("$NF")
tab = defaultdict(lambda: defaultdict(float)); calls = defaultdict(set)
for f in glob.glob("$out/a/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        n = re.sub(r"^void ", "", r["Kernel_Name"]).split("(")[0][-60:]
        tab[n][r["Counter_Name"]] += float(r["Counter_Value"]); calls[n].add(r["Dispatch_Id"])
dur = defaultdict(float)
for f in glob.glob("$out/t/*/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        n = re.sub(r"^void ", "", r["Kernel_Name"]).split("(")[0][-60:]
        dur[n] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
print("%-62s %8s %9s %10s %7s %7s %7s %9s" % ("kernel", "us/fwd", "waves/fwd", "wave us", "wait%", "istall%", "active%", "VALU busy%"))
rows = sorted(tab.items(), key=lambda kv: -dur.get(kv[0], 0))
for n, c in rows[:26]:
    wc = c["SQ_WAVE_CYCLES"]
    if wc <= 0 or dur.get(n, 0) / nf < 2: continue
    waves = c["SQ_WAVES"] / nf
    us = dur[n] / nf
    life = wc / max(c["SQ_WAVES"], 1) * 4 / 2.4e3          # quad-cycles -> us at 2.4 GHz
    valu_busy = c["SQ_ACTIVE_INST_VALU"] / nf * 4 / 1024 / (us * 2.4e3) * 100   # quad-cycles per SIMD over the kernel's cycles
    print("%-62s %8.1f %9.0f %10.2f %7.1f %7.1f %7.1f %9.1f" % (n, us, waves, life, 100 * c["SQ_WAIT_ANY"] / wc, 100 * c["SQ_WAIT_INST_ANY"] / wc, 100 * c["SQ_ACTIVE_INST_ANY"] / wc, valu_busy))
PY
cat $out/sq.txt
