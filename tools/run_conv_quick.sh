#!/bin/bash
# kernel_bench conv of base / variant libraries / base on one box (no tests, no bench): ablation builds whose outputs are wrong by design
cd "$(dirname "$0")/.."
for tag in base "$@" base; do
  echo "== $tag"
  if [ $tag = base ]; then unset FLDR_LIB; else export FLDR_LIB=tools/stamps/libfldr_$tag.so; fi
  timeout -k 10 200 python tools/kernel_bench.py conv 2>&1 | grep -E "288x 480|1152x1920|576x 960|timeouts" || exit 1
done
