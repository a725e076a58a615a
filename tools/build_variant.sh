#!/bin/bash
# tools/build_variant.sh <name> <extra compiler flags...>: libmistitch built with extra flags into tools/variants/<name>.so
set -e
name=$1; shift
mkdir -p tools/variants /tmp/variant_$name
F="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -mllvm -amdgpu-mfma-vgpr-form=1 -Wno-unused-function"
cd image-stitcher_amd/csrc
for f in api fusion; do hipcc $F "$@" -c $f.hip -o /tmp/variant_$name/$f.o; done
hipcc --offload-arch=gfx950 -shared -fPIC /tmp/variant_$name/api.o /tmp/variant_$name/fusion.o registration.o -o ../../tools/variants/$name.so
