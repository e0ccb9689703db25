#!/bin/bash
# rows per wave / pieces in flight of the small-token Linear kernel
for r in 2 1; do for u in 4 8; do
echo "== rows $r unroll $u"
MMFS_LIN_ROWS=$r MMFS_LIN_UNROLL=$u timeout 300 python tools/decode_kernels.py 1 2>&1 | grep "kernels per step\|linear_small" | cut -c1-120
done; done
