#!/bin/bash
# One GPU call: PCA tests + timings, bench A/B: parked projections (all levels / none), one-pixel splat walk.
cd "$(dirname "$0")/.."
out=gpurun_out/r3c.txt; : > $out
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "pca" > gpurun_out/t_pca.txt 2>&1 || { tail -30 gpurun_out/t_pca.txt; exit 1; }
timeout -k 10 200 python tools/kernel_bench.py pca >> $out 2>&1 || exit 1
B="python bench.py --steps 40 --warmup 5 --no-cpu-baseline --fp16-mode-steps 0 --varying-motion-steps 0 --incl-ingest-steps 0 --multi-t-pairs 0"
for cfg in "raw0 0 -" "rawnone 1099511627776 -" "raw4m 4194304 -" "noquad 0 tools/stamps/libfldr_noquad.so" "raw0 0 -" "rawnone 1099511627776 -"; do
  set -- $cfg
  export FLDR_PCA_RAW_MIN_BYTES=$2
  if [ "$3" = "-" ]; then unset FLDR_LIB; else export FLDR_LIB=$3; fi
  timeout -k 10 300 $B > gpurun_out/bench_r3c_$1.json 2>> $out || exit 1
  python - <<PY >> $out
import json
d=json.loads(open("gpurun_out/bench_r3c_$1.json").read().strip().splitlines()[-1])
print("bench $1:", d['value'], d['ms_per_step'], d['sustained']['ms_per_step'], d['config']['single_stream_latency_ms'])
PY
done
