#!/bin/bash
# One GPU call: dec3 tile order, image-splat ablations, deferred-epilogue ring A/B.
cd "$(dirname "$0")/.."
out=gpurun_out/r3b.txt; : > $out
timeout -k 10 200 python tools/kernel_bench.py dec3 >> $out 2>&1 || exit 1
echo "== splat base" >> $out; timeout -k 10 120 python tools/splat_acc64_probe.py images >> $out 2>&1 || exit 1
for tag in sab1 sab2 sab3; do
  echo "== splat $tag" >> $out
  FLDR_LIB=tools/stamps/libfldr_$tag.so timeout -k 10 120 python tools/splat_acc64_probe.py images >> $out 2>&1 || exit 1
done
MODES="conv" bash tools/run_ab.sh rdefer rdefer || exit 1
cat gpurun_out/ab.txt >> $out
