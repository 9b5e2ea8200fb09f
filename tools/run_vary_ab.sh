#!/bin/bash
# the bench's varying-motion leg (and the headline) for base / variant libraries on one box
cd "$(dirname "$0")/.."
for tag in base "$@" base; do
  if [ $tag = base ]; then unset FLDR_LIB; else export FLDR_LIB=tools/stamps/libfldr_$tag.so; fi
  timeout -k 10 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --fp16-mode-steps 0 --varying-motion-steps 120 --incl-ingest-steps 0 --multi-t-pairs 0 --sustained-s 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$tag', d['value'], d['sustained']['value'], d['varying_motion']['pairs_per_s_this_gpu'])"
done
