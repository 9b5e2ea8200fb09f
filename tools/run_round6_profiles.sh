#!/bin/bash
# Round 6: the whole profile set on ONE box, final tree (inside gpurun): GPU suite, forward PMC table + dominant conv + correlation + bench
# under rocprofv3 (prof_round.sh), wave states, launch list, dec23 issue counters, energy by stage, then the plain bench.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r6f
python -m pytest tests -m gpu -q > gpurun_out/r6f/gputests.txt 2>&1; tail -3 gpurun_out/r6f/gputests.txt
bash tools/prof_round.sh r06 > gpurun_out/r6f/prof_round.log 2>&1; tail -3 gpurun_out/r6f/prof_round.log
bash tools/prof_forward_sq.sh r06_sq > gpurun_out/r6f/sq.log 2>&1; echo "sq done"
bash tools/trace_forward_launches.sh r06_launches > gpurun_out/r6f/launches.log 2>&1; tail -1 gpurun_out/r6f/launches.log
bash tools/prof_dec23_sq.sh r06_d23sq > gpurun_out/r6f/d23sq.log 2>&1; echo "d23 counters done"
timeout -k 10 400 bash tools/energy_table.sh r06 4 > gpurun_out/r6f/energy.log 2>&1; tail -22 gpurun_out/r6f/energy.log
timeout -k 10 600 python tools/concurrency_check.py 6 > gpurun_out/r6f/concurrency.txt 2>&1; tail -1 gpurun_out/r6f/concurrency.txt
python bench.py > gpurun_out/r6f/bench.json 2> gpurun_out/r6f/bench.err; tail -c 300 gpurun_out/r6f/bench.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r6f/bench.json").read().strip().splitlines()[-1])
print("bench:", d["value"], d["ms_per_step"], d["sustained"], d["config"]["single_stream_latency_ms"], d["roofline"]["launch_ms"], d["roofline"]["frac"], d["roofline"]["path"]["frac"])
PY
