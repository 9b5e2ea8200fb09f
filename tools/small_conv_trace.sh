#!/bin/bash
# Kernel trace of tools/small_conv_probe.py for two trees / variant libraries (inside gpurun): per kernel name, mean duration and mean
# start-to-start interval inside the 10-launch graphs.   bash tools/small_conv_trace.sh <tag> <name>=<tree>[@lib] ...
tag=$1; shift
root=$GRAFT_REPO_ROOT
mkdir -p $root/gpurun_out/$tag
cd /tmp && export TMPDIR=/tmp
for spec in "$@"; do
  name=${spec%%=*}; rest=${spec#*=}; tree=${rest%%@*}; lib=""
  if [[ "$rest" == *@* ]]; then lib=${rest#*@}; fi
  if [ -n "$lib" ]; then export FLDR_LIB=$root/$lib; else unset FLDR_LIB; fi
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $root/gpurun_out/$tag/$name -- python3 $root/$tree/tools/small_conv_probe.py > $root/gpurun_out/$tag/$name.log 2>&1
done
python3 - $root/gpurun_out/$tag "$@" <<'PY'
import csv, glob, sys, collections, re
root = sys.argv[1]
for spec in sys.argv[2:]:
    name = spec.split("=")[0]
    f = glob.glob(root + "/" + name + "/*/*_kernel_trace.csv")[-1]
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    dur = collections.defaultdict(list); gap = collections.defaultdict(list)
    for a, b in zip(rows, rows[1:]):
        if "ring" not in a["Kernel_Name"]: continue
        k = re.sub(r"\(.*", "", a["Kernel_Name"])[:58] + " g" + a["Grid_Size_X"]
        dur[k].append((int(a["End_Timestamp"]) - int(a["Start_Timestamp"])) / 1e3)
        if b["Kernel_Name"] == a["Kernel_Name"] and b["Grid_Size_X"] == a["Grid_Size_X"]:
            g = (int(b["Start_Timestamp"]) - int(a["Start_Timestamp"])) / 1e3
            if g < 100: gap[k].append(g)
    print("== " + name)
    for k in dur:
        d = sorted(dur[k]); g = sorted(gap[k]) or [0]
        print("  %-72s n=%4d  duration median %6.2f   start-to-start median %6.2f" % (k, len(d), d[len(d) // 2], g[len(g) // 2]))
PY
