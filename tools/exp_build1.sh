#!/bin/bash
# One-file variant of tools/exp_build.sh: recompiles ONLY the named source with the flags -- through the same steps as
# csrc/Makefile (device assembly -> tools/fix_pk_opsel.py -> code object -> bundle -> host object; RAW=1 skips the
# rewrite: what hipcc alone makes of it) -- and links it with the objects of the regular build (run `make` first) into
# mm-interleaved_amd/csrc/build/exp/<name>.so
# usage: tools/exp_build1.sh name file(.hip, without suffix) "-DFLAG=1"
set -e
cd "$(dirname "$0")/../mm-interleaved_amd/csrc"
name=$1; file=$2; flags=$3
LLVM=/opt/rocm/lib/llvm/bin
CXX="/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wall -Wno-unused-function $flags"
o=build/exp/$name
mkdir -p $o
$CXX --cuda-device-only -S $file.hip -o $o/$file.dev.s 2> >(grep -v "hip-link" >&2)
[ -n "$RAW" ] || python3 ../../tools/fix_pk_opsel.py $o/$file.dev.s
$LLVM/clang -target amdgcn-amd-amdhsa -mcpu=gfx950 -c $o/$file.dev.s -o $o/$file.dev.o
$LLVM/lld -flavor gnu -m elf64_amdgpu --no-undefined -shared -o $o/$file.hsaco $o/$file.dev.o
$LLVM/clang-offload-bundler -type=o -bundle-align=4096 -targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950 -input=/dev/null -input=$o/$file.hsaco -output=$o/$file.hipfb
$CXX --cuda-host-only -Xclang -fcuda-include-gpubinary -Xclang $o/$file.hipfb -c $file.hip -o $o/$file.o
objs=$(ls build/*.o | grep -v "build/$file.o" | grep -v "\.dev\.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/exp/$name.so $objs $o/$file.o
echo built build/exp/$name.so
