#!/bin/bash
# A/B on ONE box: the product library against tools/stamps/libfldr_<tag>.so (kernel_bench modes given in $MODES, then bench.py).
cd "$(dirname "$0")/.."
out=gpurun_out/ab.txt; : > $out
for tag in base "$@" base; do
  echo "== $tag" >> $out
  if [ $tag = base ]; then unset FLDR_LIB; else export FLDR_LIB=tools/stamps/libfldr_$tag.so; fi
  for m in $MODES; do timeout -k 10 200 python tools/kernel_bench.py $m >> $out 2>&1 || exit 1; done
  timeout -k 10 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --fp16-mode-steps 0 --varying-motion-steps 0 --incl-ingest-steps 0 --multi-t-pairs 0 --config5-steps 0 > gpurun_out/bench_ab_$tag.json 2>> $out || exit 1
  python - <<PY >> $out
import json
d=json.loads(open("gpurun_out/bench_ab_$tag.json").read().strip().splitlines()[-1])
print("bench $tag:", d['value'], d['ms_per_step'], d['sustained']['ms_per_step'], d['config']['single_stream_latency_ms'], d['roofline']['launch_ms'])
PY
done
