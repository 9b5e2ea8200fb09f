#!/bin/bash
# Kernel trace of NF single-stream 4K forwards on the GPU box; prints the per-kernel summary of the last forward.
# usage (inside gpurun): bash tools/prof_forward.sh <tag>
set -e
tag=${1:-fwd}
root=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $root/gpurun_out/$tag -- python3 $root/tools/one_forward.py > $root/gpurun_out/$tag.log 2>&1
csv=$(ls $root/gpurun_out/$tag/*/*_kernel_trace.csv | tail -1)
python3 $root/tools/trace_timeline.py $csv > $root/gpurun_out/$tag.timeline.txt
sed -n '/by kernel/,$p' $root/gpurun_out/$tag.timeline.txt
