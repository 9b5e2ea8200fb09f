#!/bin/bash
# Saved temporaries of prep_kernels.hip for tools/asm_pad_variant.py:   asm_variant_setup.sh [-DFOO=1 ...]   (-> gpurun_out/st, build.log)
set -e
R=$(cd "$(dirname "$0")/.." && pwd); ST=$R/gpurun_out/st
mkdir -p $ST
python3 -c "import os,sys; d=sys.argv[1]; [os.remove(os.path.join(d,f)) for f in os.listdir(d)]" $ST
cd $ST
/opt/rocm/bin/hipcc -v -save-temps @$R/fldr-vfi_amd/csrc/hipcc_flags.rsp -fPIC -DFLDR_TEST_HOOKS -I$R/include -I$R/fldr-vfi_amd/csrc -Wno-unused-function "$@" \
    -c $R/fldr-vfi_amd/csrc/prep_kernels.hip -o prep_var.o 2> build.log
grep -c '^ "' build.log
