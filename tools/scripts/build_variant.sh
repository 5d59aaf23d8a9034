#!/bin/bash
# Builds jpegxl-rs_amd/lib_$AB/libjxl.so = the tree's library with kernels.hip compiled under extra flags (A/B variants for tools/scripts/ab_lib.sh).
# Usage: bash tools/scripts/build_variant.sh -DJXL_IDCT_T4=256 [...]
set -e
cd "$(dirname "$0")/../../jpegxl-rs_amd"
make -s -j8
AB=${AB_DIR:-ab}; mkdir -p build_$AB lib_$AB
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fexceptions -DJXL_NT16 -Wno-unused-function -Wno-unused-result "$@" -c csrc/kernels.hip -o build_$AB/kernels.o
OBJ="build_$AB/kernels.o build/kernels_features.o build/host_features.o build/jpeg_recon.o build/host_parse.o build/decoder.o build/pipeline.o build/scheduler.o build/gather.o build/jxl_abi.o build/jxl_stubs.o build/icc_profile.o"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o lib_$AB/libjxl.so $OBJ -ldl -lpthread -Wl,-soname,libjxl.so.0.11
ls -la lib_$AB/libjxl.so
