#!/bin/bash
# Round profile set (run inside gpurun): per-kernel forward table with PMC traffic, the dominant conv alone (kernel stats,
# FETCH / WRITE / SQ+GRBM passes), the stand-alone correlation operator, the bench command under rocprofv3 --stats.
#   bash tools/prof_round.sh <tag>      -> gpurun_out/<tag>/...
set -e
tag=${1:-r02_final}
root=$GRAFT_REPO_ROOT
out=$root/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
bash $root/tools/prof_forward_pmc.sh $tag/fwd_pmc > $out/fwd_pmc.log 2>&1 || true
echo "forward pmc done"
export REPS=20
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/conv96_trace -- python3 $root/tools/one_conv_spk.py > /dev/null 2>&1
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/conv96_fetch -- python3 $root/tools/one_conv_spk.py > /dev/null 2>&1
timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/conv96_write -- python3 $root/tools/one_conv_spk.py > /dev/null 2>&1
timeout -k 10 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $out/conv96_sq -- python3 $root/tools/one_conv_spk.py > /dev/null 2>&1
echo "conv96 done"
export REPS=5
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/corr_trace -- python3 $root/tools/one_corr.py > /dev/null 2>&1
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/corr_fetch -- python3 $root/tools/one_corr.py > /dev/null 2>&1
timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/corr_write -- python3 $root/tools/one_corr.py > /dev/null 2>&1
echo "correlation done"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/bench_trace -- python3 $root/bench.py --no-cpu-baseline --steps 20 --warmup 3 --sustained-s 0 > $out/bench_under_rocprof.json 2> $out/bench_under_rocprof.err
echo "bench trace done"
