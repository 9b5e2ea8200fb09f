#!/bin/bash
# A/B of environment switches on one box: ENVS="NAME=0 NAME=1 ..." (each run once per ROUNDS)
cd "$(dirname "$0")/.."
out=gpurun_out/ab_env.txt; : > $out
B="python bench.py --steps 40 --warmup 5 --no-cpu-baseline --fp16-mode-steps 0 --varying-motion-steps 0 --incl-ingest-steps 0 --multi-t-pairs 0"
for r in $(seq 1 ${ROUNDS:-2}); do for e in $ENVS; do
  env $e timeout -k 10 300 $B > gpurun_out/bench_ab_env.json 2>> $out || exit 1
  python - <<PY >> $out
import json
d=json.loads(open("gpurun_out/bench_ab_env.json").read().strip().splitlines()[-1])
print("bench $e:", d['value'], d['ms_per_step'], d['sustained']['ms_per_step'], d['config']['single_stream_latency_ms'])
PY
done; done
