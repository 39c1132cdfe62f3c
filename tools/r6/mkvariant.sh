#!/bin/bash
# build a variant of the library with ONE source recompiled with extra flags: tools/r6/mkvariant.sh NAME source.hip "-DX=1 ..."
set -e
cd "$(dirname "$0")/../.."
N=$1; SRC=$2; FL=$3
mkdir -p change3d_amd/lib/obj_var
O=change3d_amd/lib/obj_var/${N}_$(basename $SRC .hip).o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-strict-aliasing -Wno-comment -Wno-unused-result $FL -c change3d_amd/csrc/$SRC -o $O
OBJS=$(ls change3d_amd/lib/obj/*.o | grep -v "_clk.o" | grep -v "/$(basename $SRC .hip).o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o change3d_amd/lib/libchange3d_hip_$N.so $OBJS $O
echo built change3d_amd/lib/libchange3d_hip_$N.so
