#!/bin/bash
# Clock / power of the GPU while the bench loop runs (GPU box): is the step power-limited?  -> gpurun_out/power_probe.txt
cd "$(dirname "$0")/.."
out=gpurun_out/power_probe.txt; : > $out
python bench.py --steps 40 --warmup 5 --no-cpu-baseline --fp16-mode-steps 0 --varying-motion-steps 0 --incl-ingest-steps 0 --multi-t-pairs 0 --sustained-s ${SUST:-12} > gpurun_out/power_probe_bench.json 2>/dev/null &
pid=$!
sleep ${DELAY:-14}
for i in $(seq 1 16); do
  /opt/rocm/bin/rocm-smi --showclocks --showpower --showuse 2>/dev/null | grep -E "sclk|mclk|Power|busy|Socket" >> $out
  echo "--" >> $out
  sleep 0.5
done
wait $pid
python -c "import json; d=json.loads(open('gpurun_out/power_probe_bench.json').read().strip().splitlines()[-1]); print('bench', d['value'], d['sustained'])" >> $out
tail -60 $out
