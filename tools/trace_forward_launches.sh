#!/bin/bash
# Every launch of one warm 4K forward in order: kernel, grid (workgroups), duration and the gap to the previous launch's end.
# usage (inside gpurun): bash tools/trace_forward_launches.sh <tag>   -> gpurun_out/<tag>/launches.txt
tag=${1:-fwd_launches}
export FW=3840 FH=2160 NF=${NF:-3}
root=$GRAFT_REPO_ROOT
out=$root/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $out/t -- python3 $root/tools/one_forward.py > $out/t.log 2>&1 || exit 1
python3 - <<PY > $out/launches.txt
import csv, glob, re
rows = []
for f in glob.glob("$out/t/*/*kernel_trace.csv"):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last forward = the last 60-61 launches ending with the synthesis kernel
names = [re.sub(r"^void ", "", r["Kernel_Name"]).split("(")[0] for r in rows]
last = max(i for i, n in enumerate(names) if "dec23_synth" in n or "dec3_synth" in n)
first = max(i for i, n in enumerate(names[:last]) if "dec23_synth" in n or "dec3_synth" in n) + 1
prev_end = None
tot = 0.0
for r, n in list(zip(rows, names))[first:last + 1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    wg = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"]) // max(1, int(r["Workgroup_Size_X"]) * int(r["Workgroup_Size_Y"]) * int(r["Workgroup_Size_Z"]))
    gap = (s - prev_end) / 1e3 if prev_end else 0.0
    print("%-70s wgs %6d  lds %6s  %8.2f us  gap %6.2f" % (n[-70:], wg, r.get("LDS_Block_Size", "?"), (e - s) / 1e3, gap))
    prev_end = e; tot += (e - s) / 1e3
print("sum of durations %.1f us over %d launches" % (tot, last + 1 - first))
PY
cat $out/launches.txt
