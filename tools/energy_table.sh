#!/bin/bash
# Energy of one 4K forward by stage (GPU box): tools/energy_by_stage.py loops every stage of the forward for SECS seconds each while this
# script samples board power and shader clock (rocm-smi) with time stamps; the join writes gpurun_out/<tag>_energy_by_kernel.{txt,json}.
#   bash tools/energy_table.sh <tag> [SECS]
cd "$(dirname "$0")/.."
tag=${1:-r06}; secs=${2:-4}
mkdir -p gpurun_out
st=gpurun_out/${tag}_energy_stages.log; sm=gpurun_out/${tag}_energy_samples.log
: > $st; : > $sm
/opt/rocm/bin/rocm-smi --showmaxpower 2>/dev/null | grep -i "power" > gpurun_out/${tag}_energy_cap.txt
python tools/energy_by_stage.py run $secs > $st 2> gpurun_out/${tag}_energy_stages.err &
pid=$!
while kill -0 $pid 2>/dev/null; do
  s=$(/opt/rocm/bin/rocm-smi --showclocks --showpower 2>/dev/null)
  w=$(echo "$s" | grep -o 'Power (W): [0-9.]*' | grep -o '[0-9.]*$' | head -1)
  c=$(echo "$s" | grep 'sclk' | grep -o '([0-9]*Mhz)' | tr -d '()Mhz' | head -1)
  echo "$(date +%s.%N) ${w:-0} ${c:-0}" >> $sm
done
wait $pid; rc=$?
tail -3 gpurun_out/${tag}_energy_stages.err
python tools/energy_by_stage.py join $st $sm gpurun_out/${tag}_energy_by_kernel.txt gpurun_out/${tag}_energy_by_kernel.json
cat gpurun_out/${tag}_energy_cap.txt
exit $rc
