#!/bin/bash
# Board power and shader clock while ONE kernel of the 4K forward runs back to back (GPU box): which kernels run at the power limit?
#   bash tools/power_by_kernel.sh [kernel ...]     -> gpurun_out/power_by_kernel.txt
cd "$(dirname "$0")/.."
out=gpurun_out/power_by_kernel.txt; : > $out
for k in ${@:-idle conv96 conv_dec2 prep enc1 dec3 splat pca}; do
  python tools/kernel_loop.py $k ${SECS:-5} > gpurun_out/kernel_loop.log 2>/dev/null &
  pid=$!
  for w8 in $(seq 1 600); do grep -q START gpurun_out/kernel_loop.log 2>/dev/null && break; sleep 0.25; done     # model load + warm-up
  sleep 1.5
  w=""; c=""
  for i in 1 2 3 4; do
    s=$(/opt/rocm/bin/rocm-smi --showclocks --showpower 2>/dev/null)
    w="$w $(echo "$s" | grep -o 'Power (W): [0-9.]*' | grep -o '[0-9.]*$')"
    c="$c $(echo "$s" | grep 'sclk' | grep -o '([0-9]*Mhz)' | tr -d '()Mhz')"
    sleep 0.4
  done
  wait $pid
  echo "$k: power W [$w ] sclk MHz [$c ]  $(grep END gpurun_out/kernel_loop.log | cut -d' ' -f4-)" >> $out
done
cat $out
