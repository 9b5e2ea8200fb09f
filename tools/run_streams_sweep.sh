#!/bin/bash
# the 40-step bench at several (streams, pairs) settings on one box
cd "$(dirname "$0")/.."
out=gpurun_out/streams.txt; : > $out
for sp in "2 4" "3 4" "3 6" "4 4" "4 8" "6 6"; do set -- $sp
  timeout -k 10 300 python bench.py --steps 48 --warmup 6 --streams $1 --pairs $2 --no-cpu-baseline --fp16-mode-steps 0 --varying-motion-steps 0 --incl-ingest-steps 0 --multi-t-pairs 0 > gpurun_out/bench_streams.json 2>> $out || exit 1
  python - <<PY >> $out
import json
d=json.loads(open("gpurun_out/bench_streams.json").read().strip().splitlines()[-1])
print("streams $1 pairs $2:", d['value'], d['ms_per_step'], d['sustained']['ms_per_step'])
PY
done
cat $out | grep streams
