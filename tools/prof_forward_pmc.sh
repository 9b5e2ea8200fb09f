#!/bin/bash
# Per-kernel duration + HBM-side traffic of single-stream 4K forwards on the GPU box (run inside gpurun):
#   pass 1  rocprofv3 --kernel-trace --stats          -> durations
#   pass 2  rocprofv3 --pmc FETCH_SIZE  (own run)     -> read bytes   (MI355X_MICROARCH.md: separate --pmc passes)
#   pass 3  rocprofv3 --pmc WRITE_SIZE  (own run)     -> written bytes
# plus the same two PMC passes over tools/ubench/plane_bw_bench (known byte counts at 4 B and 16 B per lane) to calibrate
# the gfx950 FETCH_SIZE correction for the access widths these kernels use.
# usage: bash tools/prof_forward_pmc.sh <tag> [W H]     -> gpurun_out/<tag>/{summary.json,summary.txt}
set -e
tag=${1:-r02_forward}
export FW=${2:-3840} FH=${3:-2160} NF=${NF:-3}
root=$GRAFT_REPO_ROOT
out=$root/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 $root/tools/one_forward.py > $out/trace.log 2>&1
echo "trace done"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -- python3 $root/tools/one_forward.py > $out/fetch.log 2>&1
echo "fetch done"
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write -- python3 $root/tools/one_forward.py > $out/write.log 2>&1
echo "write done"
if [ -x $root/tools/ubench/plane_bw_bench ]; then
  timeout -k 10 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/cal_fetch -- $root/tools/ubench/plane_bw_bench > $out/cal_fetch.log 2>&1
  timeout -k 10 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/cal_write -- $root/tools/ubench/plane_bw_bench > $out/cal_write.log 2>&1
  echo "calibration done"
fi
python3 $root/tools/pmc_forward_summary.py $out $NF > $out/summary.txt
cat $out/summary.txt
