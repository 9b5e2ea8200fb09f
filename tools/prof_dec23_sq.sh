#!/bin/bash
# Instruction-issue counters of the fused dec2 -> dec3 kernel (and of the two-kernel path beside it): what the SIMDs of a CU spend a tile on.
# usage (inside gpurun): bash tools/prof_dec23_sq.sh <tag>   -> gpurun_out/<tag>/sq.txt
tag=${1:-d23_sq}
root=$GRAFT_REPO_ROOT
out=$root/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_WAVES SQ_INSTS_BRANCH --output-format csv -d $out/a -- python3 $root/tools/dec23_probe.py > $out/a.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES --output-format csv -d $out/b -- python3 $root/tools/dec23_probe.py > $out/b.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_INST_CYCLES_VMEM SQ_LDS_IDX_ACTIVE --output-format csv -d $out/c -- python3 $root/tools/dec23_probe.py > $out/c.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $out/t -- python3 $root/tools/dec23_probe.py > $out/t.log 2>&1 || exit 1
python3 - <<PY > $out/sq.txt
import csv, glob, re
from collections import defaultdict
tab = defaultdict(lambda: defaultdict(float)); calls = defaultdict(set)
for d in "abc":
    for f in glob.glob("$out/%s/*/*counter_collection.csv" % d):
        for r in csv.DictReader(open(f)):
            n = re.sub(r"^void ", "", r["Kernel_Name"]).split("(")[0][-44:]
            tab[n][r["Counter_Name"]] += float(r["Counter_Value"]); calls[(n, d)].add(r["Dispatch_Id"])
dur = defaultdict(float); cnt = defaultdict(int)
for f in glob.glob("$out/t/*/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        n = re.sub(r"^void ", "", r["Kernel_Name"]).split("(")[0][-44:]
        dur[n] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3; cnt[n] += 1
for n in sorted(tab, key=lambda k: -dur.get(k, 0))[:4]:
    c = tab[n]; nc = {d: max(1, len(calls[(n, d)])) for d in "abc"}
    us = dur[n] / max(1, cnt[n])
    print("%s: %.1f us per launch (%d launches traced)" % (n, us, cnt[n]))
    for d, names in (("a", ("SQ_INSTS_VALU", "SQ_INSTS_MFMA", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM", "SQ_INSTS_SMEM", "SQ_INSTS_BRANCH", "SQ_WAVES")),
                     ("b", ("SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_LDS", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", "SQ_LDS_BANK_CONFLICT", "SQ_ACTIVE_INST_ANY", "SQ_WAVE_CYCLES")),
                     ("c", ("SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_TRANS_F64", "SQ_INSTS_VALU_CVT", "SQ_INSTS_VALU_INT32", "SQ_INST_CYCLES_VMEM", "SQ_LDS_IDX_ACTIVE"))):
        print("   " + "  ".join("%s %.3g" % (k.replace("SQ_", ""), c[k] / nc[d]) for k in names))
PY
cat $out/sq.txt
