#!/bin/bash
# generic GPU step: [TESTS_K="pytest -k expression"] tests, kernel_bench modes in $MODES, then optional bench ($BENCH=1)
cd "$(dirname "$0")/.."
out=gpurun_out/step.txt; : > $out
if [ -n "$TESTS_K" ]; then timeout -k 10 900 python -m pytest tests -x -q -m gpu -k "$TESTS_K" > gpurun_out/t_step.txt 2>&1 || { tail -30 gpurun_out/t_step.txt; exit 1; }; fi
for m in $MODES; do timeout -k 10 300 python tools/kernel_bench.py $m >> $out 2>&1 || exit 1; done
if [ -n "$BENCH" ]; then
  timeout -k 10 400 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --fp16-mode-steps 0 --varying-motion-steps 0 --incl-ingest-steps 0 --multi-t-pairs 0 > gpurun_out/bench_step.json 2>> $out || exit 1
  python - <<PY >> $out
import json
d=json.loads(open("gpurun_out/bench_step.json").read().strip().splitlines()[-1])
print("bench:", d['value'], d['ms_per_step'], d['sustained']['ms_per_step'], d['config']['single_stream_latency_ms'])
PY
fi
