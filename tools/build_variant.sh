#!/bin/bash
# Builds a variant of the library with extra compiler flags for ONE source file (the other objects are
# the in-tree build's):  tools/build_variant.sh <name> <source.hip> <flags...>   ->  bin_ab/lib<name>.so
set -e
name=$1; src=$2; shift 2
cd "$(dirname "$0")/../medical_image_segmentation_amd/csrc"
mkdir -p ../../bin_ab build
obj=../../bin_ab/${name}_${src%.hip}.o
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -Wno-unused-variable "$@" -c $src -o $obj
objs=""
for f in core loss_metrics pool_upsample bn conv_direct conv_api conv_mfma optim slide amsfn rdam; do
  if [ "$f.hip" = "$src" ]; then objs="$objs $obj"; else objs="$objs build/$f.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o ../../bin_ab/lib${name}.so $objs
echo built bin_ab/lib${name}.so
