#!/bin/bash
# grad_loc / grad_attn: the fused LDS kernel (MMFS_TAPS_ALGO=mma) against the default routing, per workload
for w in "$@"; do
 for a in "" mma; do
  echo "== $w taps=${a:-default}"
  MMFS_TAPS_ALGO=$a python bench.py --no-cpu-baseline --steps 30 --warmup 5 --workload $w 2>/dev/null | grep '^{' | python -c "
import sys, json
for l in sys.stdin:
    r = json.loads(l); print('  ms/step', r['ms_per_step'], r['kernels_mean_us'])"
 done
done
