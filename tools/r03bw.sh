#!/bin/bash
for rep in 1 2; do for e in 1 0; do
echo "== early $e"; MMFS_LIN_EARLY=$e timeout 300 python tools/decode_kernels.py 1 2>&1 | grep "kernels per step\|linear_small" | cut -c1-120
done; done
