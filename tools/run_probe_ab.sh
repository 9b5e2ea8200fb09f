#!/bin/bash
# a probe script ($PROBE) on the product library against tools/stamps/libfldr_<tag>.so variants ($TAGS) on one box
cd "$(dirname "$0")/.."
for tag in base ${TAGS} base; do
  echo "== $tag"
  if [ $tag = base ]; then unset FLDR_LIB; else export FLDR_LIB=tools/stamps/libfldr_$tag.so; fi
  timeout -k 10 300 python $PROBE 2>&1 | grep -v amdgpu.ids || exit 1
done
