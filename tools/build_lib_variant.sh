#!/bin/bash
# Whole-library A/B build with extra -D flags:  build_lib_variant.sh <tag> [-DFOO ...]  ->  tools/stamps/libfldr_<tag>.so
# (FLDR_LIB=<that path> selects it in fldr_hip.py / bench.py / the tools)
set -e
R=$(cd "$(dirname "$0")/.." && pwd); tag=$1; shift
B=/tmp/fldr_variant_$tag; rm -rf $B; mkdir -p $B; cp $R/fldr-vfi_amd/csrc/*.hip $R/fldr-vfi_amd/csrc/*.h $B/
FL="@$R/fldr-vfi_amd/csrc/hipcc_flags.rsp -fPIC -DFLDR_TEST_HOOKS -I$R/include -I$B -Wno-unused-function"
for f in $B/*.hip; do /opt/rocm/bin/hipcc $FL "$@" -c $f -o ${f%.hip}.o & done; wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/tools/stamps/libfldr_$tag.so $B/*.o
