#!/bin/bash
# Ring-conv variants on ONE box (tools/stamps/build_variant.sh conv_ring_kernels.hip <tag> -D...): the conv bit-identity tests on every
# variant library, then per-layer timings (kernel_bench conv) of base / variants / base, then the 40-step bench of each.
#   bash tools/run_conv_ab.sh <tag> ...        BENCH=0 skips the bench
cd "$(dirname "$0")/.."
out=gpurun_out/conv_ab.txt; : > $out
for tag in "$@"; do
  echo "== tests $tag" >> $out
  FLDR_LIB=tools/stamps/libfldr_$tag.so timeout -k 10 400 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "spk_conv_bit_identical or multi_level_conv or conv_matches_torch or split_fp16_conv_is_fp32 or range_guard or model_matches" >> $out 2>&1 || { tail -30 $out; exit 1; }
done
for tag in base "$@" base; do
  echo "== $tag" >> $out
  if [ $tag = base ]; then unset FLDR_LIB; else export FLDR_LIB=tools/stamps/libfldr_$tag.so; fi
  timeout -k 10 200 python tools/kernel_bench.py conv 2>&1 | grep -v amdgpu.ids >> $out || exit 1
done
if [ "${BENCH:-1}" = 1 ]; then
for tag in base "$@" base; do
  if [ $tag = base ]; then unset FLDR_LIB; else export FLDR_LIB=tools/stamps/libfldr_$tag.so; fi
  timeout -k 10 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --fp16-mode-steps 0 --varying-motion-steps 0 --incl-ingest-steps 0 --multi-t-pairs 0 > gpurun_out/bench_ab_$tag.json 2>> $out || exit 1
  python - <<PY >> $out
import json
d=json.loads(open("gpurun_out/bench_ab_$tag.json").read().strip().splitlines()[-1])
print("bench $tag:", d['value'], d['ms_per_step'], d['sustained']['ms_per_step'], d['config']['single_stream_latency_ms'], d['roofline']['launch_ms'])
PY
done
fi
cat $out | grep -v "^$" | tail -120
