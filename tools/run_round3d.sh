#!/bin/bash
# plane streaming rate by access width; multi-t test + bench record with side streams
cd "$(dirname "$0")/.."
out=gpurun_out/r3d.txt; : > $out
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -o /tmp/plane_bw_bench tools/ubench/plane_bw_bench.hip >> $out 2>&1 || exit 1
timeout -k 10 120 /tmp/plane_bw_bench >> $out 2>&1 || exit 1
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "multi_t or pca" > gpurun_out/t_mt.txt 2>&1 || { tail -30 gpurun_out/t_mt.txt; exit 1; }
timeout -k 10 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --fp16-mode-steps 0 --varying-motion-steps 0 --incl-ingest-steps 0 --multi-t-pairs 6 > gpurun_out/bench_r3d.json 2>> $out || exit 1
python - <<PY >> $out
import json
d=json.loads(open("gpurun_out/bench_r3d.json").read().strip().splitlines()[-1])
print("bench:", d['value'], d['ms_per_step'], d['sustained']['ms_per_step'], d['config']['single_stream_latency_ms'])
print(json.dumps(d['multi_t'], indent=1))
PY
