#!/bin/bash
# Diagnostic / A-B builds of the library (CPU container; the .so travels to the GPU box under build/, which is git-ignored):
#   scripts/build_variant.sh <name> <file.hip> [extra hipcc flags ...]     -> build/libtransgo_hip_<name>.so
# Only <file.hip> is recompiled with the extra flags; the other objects are the product build's (run `make -C transgo_amd/csrc` first).
set -e
cd "$(dirname "$0")/../transgo_amd/csrc"
name=$1; src=$2; shift 2
mkdir -p ../../build
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wall -Wno-unused-function "$@" -c $src -o ../../build/${src%.hip}_$name.o
objs=""
for o in env.o engine.o net.o replay.o rng_host.o compat.o; do
  if [ $o = ${src%.hip}.o ]; then objs="$objs ../../build/${src%.hip}_$name.o"; else objs="$objs $o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../build/libtransgo_hip_$name.so $objs
ls -la ../../build/libtransgo_hip_$name.so
