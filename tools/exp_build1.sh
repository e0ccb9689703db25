#!/bin/bash
# One-file variant of tools/exp_build.sh: recompiles ONLY the named source with the flags and links it with the objects of
# the regular build (run `make` first) into mm-interleaved_amd/csrc/build/exp/<name>.so
# usage: tools/exp_build1.sh name file(.hip, without suffix) "-DFLAG=1"
set -e
cd "$(dirname "$0")/../mm-interleaved_amd/csrc"
name=$1; file=$2; flags=$3
mkdir -p build/exp/$name
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wall -Wno-unused-function $flags -c $file.hip -o build/exp/$name/$file.o
objs=$(ls build/*.o | grep -v "build/$file.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/exp/$name.so $objs build/exp/$name/$file.o
echo built build/exp/$name.so
