#!/bin/bash
# tools/build_reg_variant.sh <name> <extra compiler flags...>: libmistitch with registration.hip built with extra
# flags (e.g. -DMS_STAMPS, -DMS_LONG_NF=2) into tools/variants/<name>.so; load it with MISTITCH_LIB=<path>.
set -e
name=$1; shift
mkdir -p tools/variants /tmp/variant_$name
F="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -mllvm -amdgpu-mfma-vgpr-form=1 -Wno-unused-function"
cd image-stitcher_amd/csrc
hipcc $F "$@" -c registration.hip -o /tmp/variant_$name/registration.o
hipcc $F "$@" -c api.hip -o /tmp/variant_$name/api.o
hipcc --offload-arch=gfx950 -shared -fPIC /tmp/variant_$name/api.o fusion.o /tmp/variant_$name/registration.o -o ../../tools/variants/$name.so
