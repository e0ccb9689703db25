#!/bin/bash
# The forward alone (tools/fwd_repeat.py: back to back, other locations, other VALUES) per experimental library variant.
# usage: tools/exp_fwd.sh "variant1 variant2 ..."   (variants: csrc/build/exp/<name>.so; "base" = the product library)
cd "$(dirname "$0")/.."
for v in base $1; do
  if [ "$v" = base ]; then lib=mm-interleaved_amd/libmmfs_msda.so; else lib=mm-interleaved_amd/csrc/build/exp/$v.so; fi
  echo "== $v"
  MMFS_MSDA_LIB=$PWD/$lib timeout 120 python tools/fwd_repeat.py 2>&1 | grep "^forward" | sed 's/^/  /'
done
