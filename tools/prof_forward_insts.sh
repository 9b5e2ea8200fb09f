#!/bin/bash
# Instruction mix of single-stream 4K forwards per kernel (the step runs at the board's power limit: instructions are energy):
#   rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS  (own run)  and  --pmc SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR  (own run)
# usage (inside gpurun): bash tools/prof_forward_insts.sh <tag>   -> gpurun_out/<tag>/insts.txt
set -e
tag=${1:-insts}
export FW=3840 FH=2160 NF=${NF:-3}
root=$GRAFT_REPO_ROOT
out=$root/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d $out/a -- python3 $root/tools/one_forward.py > $out/a.log 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $out/b -- python3 $root/tools/one_forward.py > $out/b.log 2>&1
python3 - <<PY > $out/insts.txt
import csv, glob, re
from collections import OrderedDict, defaultdict
nf = float("$NF")
tab = defaultdict(lambda: defaultdict(float)); calls = defaultdict(set)
for sub in ("a", "b"):
    for f in glob.glob("$out/%s/*/*counter_collection.csv" % sub):
        for r in csv.DictReader(open(f)):
            n = re.sub(r"^void ", "", r["Kernel_Name"]).split("(")[0][-64:]
            tab[n][r["Counter_Name"]] += float(r["Counter_Value"]); calls[n].add((sub, r["Dispatch_Id"]))
cols = ["SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VALU_MFMA_MOPS_F16", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR"]
print("%-66s %10s %10s %10s %12s %10s %10s   (M wave-instructions per forward)" % ("kernel", "VALU", "SALU", "LDS", "MFMA MOPS", "VMEM rd", "VMEM wr"))
rows = sorted(tab.items(), key=lambda kv: -kv[1].get("SQ_INSTS_VALU", 0))
tot = defaultdict(float)
for n, c in rows[:28]:
    print("%-66s " % n + " ".join("%10.2f" % (c.get(k, 0) / nf / 1e6) if k != "SQ_INSTS_VALU_MFMA_MOPS_F16" else "%12.2f" % (c.get(k, 0) / nf / 1e6) for k in cols))
for n, c in rows:
    for k in cols: tot[k] += c.get(k, 0) / nf / 1e6
print("%-66s " % "TOTAL" + " ".join("%10.2f" % tot[k] if k != "SQ_INSTS_VALU_MFMA_MOPS_F16" else "%12.2f" % tot[k] for k in cols))
PY
cat $out/insts.txt
