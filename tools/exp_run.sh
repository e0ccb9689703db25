#!/bin/bash
# Runs bench.py once per experimental library variant (MMFS_MSDA_LIB selects the .so).
# usage: tools/exp_run.sh "variant1 variant2 ..." [extra bench args]
cd "$(dirname "$0")/.."
variants=$1; shift
for v in base $variants; do
  if [ "$v" = base ]; then lib=mm-interleaved_amd/libmmfs_msda.so; else lib=mm-interleaved_amd/csrc/build/exp/$v.so; fi
  echo "== $v"
  MMFS_MSDA_LIB=$PWD/$lib timeout 120 python bench.py --no-cpu-baseline --steps 30 --warmup 5 "$@" 2>&1 | grep '^{' | python -c "
import sys, json
for l in sys.stdin:
    r = json.loads(l); print('  ms/step', r['ms_per_step'], r['kernels_mean_us'])"
done
