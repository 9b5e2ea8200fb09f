#!/bin/bash
# Ring-conv variants (tools/stamps/build_variant.sh conv_ring_kernels.hip <tag> -D...): per-layer timings and the bench (GPU box).
cd "$(dirname "$0")/.."
out=gpurun_out/conv_variants.txt; : > $out
timeout -k 10 600 python -m pytest tests -x -q -m gpu > gpurun_out/t_all.txt 2>&1 || exit 1
for tag in base "$@"; do
  echo "== $tag" >> $out
  if [ $tag = base ]; then unset FLDR_LIB; else export FLDR_LIB=tools/stamps/libfldr_$tag.so; fi
  timeout -k 10 200 python tools/kernel_bench.py conv_cold >> $out 2>&1 || exit 1
  timeout -k 10 200 python tools/kernel_bench.py conv >> $out 2>&1 || exit 1
  timeout -k 10 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --fp16-mode-steps 0 --varying-motion-steps 0 --incl-ingest-steps 0 --multi-t-pairs 0 > gpurun_out/bench_$tag.json 2>> $out || exit 1
done
