#!/bin/bash
# One-file A/B variant of the library: recompile a single source with extra flags and relink with the other objects.
# usage: tools/variant_obj.sh <name> <source.hip> "<extra flags>"   ->  pointvs_amd/libpvs_egnn_<name>.so
set -e
name=$1; src=$2; extra=$3
cd "$(dirname "$0")/../pointvs_amd/csrc"
mkdir -p abl_$name
x=""; [ "$src" = edge_mfma_fwd.hip ] && x="-fno-slp-vectorize -mllvm -amdgpu-mfma-vgpr-form=1"
/opt/rocm/bin/hipcc -O3 -fPIC -std=c++17 --offload-arch=gfx950 -Wno-unused-value $x $extra -c $src -o abl_$name/${src%.hip}.o
objs=""
for o in *.o; do [ "$o" = "${src%.hip}.o" ] && objs="$objs abl_$name/$o" || objs="$objs $o"; done
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $objs -o ../libpvs_egnn_$name.so
echo built ../libpvs_egnn_$name.so
