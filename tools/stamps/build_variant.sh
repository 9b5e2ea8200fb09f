#!/bin/bash
# Experimental build of ONE csrc file with extra -D flags, linked against the objects of the test build (make hooks):
#   build_variant.sh <file.hip> <tag> [-DFOO=1 ...]  ->  tools/stamps/libfldr_<tag>.so
set -e
R=$(cd "$(dirname "$0")" && pwd); C=$R/../../fldr-vfi_amd/csrc
f=$1; tag=$2; shift 2
FL="@$C/hipcc_flags.rsp -fPIC -DFLDR_TEST_HOOKS -I$R/../../include -I$C -Wno-unused-function"
base=$(basename $f .hip)
/opt/rocm/bin/hipcc $FL "$@" -c $C/$base.hip -o $R/${base}_$tag.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/libfldr_$tag.so $R/${base}_$tag.o $(ls $C/*.t.o | grep -v "/$base.t.o")
