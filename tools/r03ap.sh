#!/bin/bash
echo "== norm tests"; timeout 600 python -m pytest tests/test_modules_gpu.py -q -x -k "rmsnorm or norm or schedule or llama" 2>&1 | tail -2
for g in 1024 512 256 2048; do echo "== partial rows $g"; MMFS_NORM_BWD_GRID=$g timeout 120 python tools/norm_bench.py 2>&1 | grep rows; done
