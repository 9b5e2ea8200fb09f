#!/bin/bash
# kernel_bench modes ($MODES) of the product library against tools/stamps/libfldr_<tag>.so variants ($TAGS) on one box
cd "$(dirname "$0")/.."
for tag in base ${TAGS} base; do
  echo "== $tag"
  if [ $tag = base ]; then unset FLDR_LIB; else export FLDR_LIB=tools/stamps/libfldr_$tag.so; fi
  for m in $MODES; do timeout -k 10 200 python tools/kernel_bench.py $m 2>&1 | grep -v amdgpu.ids || exit 1; done
done
