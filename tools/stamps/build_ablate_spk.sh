#!/bin/bash
# Diagnostic builds of the split-packed conv with one phase compiled out (SPK_ABLATE=n): results are WRONG by design,
# only the timing is used (tools/stamps/run_ablate_spk.py sabN).
set -e
R=$(cd "$(dirname "$0")" && pwd); C=$R/../../fldr-vfi_amd/csrc
FL="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -I$R/../../include -Wno-unused-function"
OTHERS=$(ls $C/*.o | grep -v conv_spk_kernels.o)
for n in "$@"; do
  /opt/rocm/bin/hipcc $FL -DSPK_ABLATE=$n -c $C/conv_spk_kernels.hip -o $R/spk_ab$n.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/libfldr_sab$n.so $R/spk_ab$n.o $OTHERS
done
