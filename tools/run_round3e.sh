#!/bin/bash
cd "$(dirname "$0")/.."
out=gpurun_out/r3e.txt; : > $out
for tag in base d3a1 d3a2 d3a3 d3a4 base; do
  echo "== $tag" >> $out
  if [ $tag = base ]; then unset FLDR_LIB; else export FLDR_LIB=tools/stamps/libfldr_$tag.so; fi
  timeout -k 10 120 python tools/kernel_bench.py dec3 2>&1 | grep "tile order" >> $out || exit 1
done
