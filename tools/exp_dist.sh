#!/bin/bash
# tools/fwd_dist.py per experimental library variant.  usage: tools/exp_dist.sh "variant ..." [workload]
cd "$(dirname "$0")/.."
for v in base $1; do
  if [ "$v" = base ]; then lib=mm-interleaved_amd/libmmfs_msda.so; else lib=mm-interleaved_amd/csrc/build/exp/$v.so; fi
  echo "== $v"; MMFS_MSDA_LIB=$PWD/$lib timeout 120 python tools/fwd_dist.py $2 2>&1 | grep "^cfg\|^ref" | sed 's/^/  /'
done
