#!/bin/bash
# Timings of the acc64 splat variants built with tools/stamps/build_variant.sh (GPU box).
cd "$(dirname "$0")/.."
out=gpurun_out/acc64_variants.txt; : > $out
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "acc64 or model_ or splat" > gpurun_out/t_acc64.txt 2>&1 || exit 1
echo "== base" >> $out; timeout -k 10 120 python tools/splat_acc64_probe.py >> $out 2>&1 || exit 1
for tag in "$@"; do
  echo "== $tag" >> $out
  FLDR_LIB=tools/stamps/libfldr_$tag.so timeout -k 10 120 python tools/splat_acc64_probe.py >> $out 2>&1 || exit 1
done
python bench.py --steps 20 --warmup 5 > gpurun_out/bench_acc64v2.json 2> gpurun_out/bench_acc64v2.err || exit 1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_acc64 -o p -- python3 $GRAFT_REPO_ROOT/tools/splat_acc64_probe.py > $GRAFT_REPO_ROOT/gpurun_out/prof_acc64.log 2>&1
