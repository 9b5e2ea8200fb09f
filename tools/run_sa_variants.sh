#!/bin/bash
# acc64 splat configurations (tools/stamps/libfldr_<tag>.so) on one box: [tests, then] the probe per variant
cd "$(dirname "$0")/.."
out=gpurun_out/sa_variants.txt; : > $out
if [ -z "$NOTEST" ]; then
timeout -k 10 400 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "acc64 or splat or model_" > gpurun_out/t_acc64.txt 2>&1 || { tail -30 gpurun_out/t_acc64.txt; exit 1; }
fi
for tag in base "$@" base; do
  echo "== $tag" >> $out
  if [ $tag = base ]; then unset FLDR_LIB; else export FLDR_LIB=tools/stamps/libfldr_$tag.so; fi
  timeout -k 10 120 python tools/splat_acc64_probe.py ${WHICH:-all} >> $out 2>&1 || exit 1
done
