#!/bin/bash
for g in 1024 512 256 2048; do echo "== partial rows $g"; MMFS_NORM_BWD_GRID=$g timeout 120 python tools/norm_bench.py 2>&1 | grep -v "amdgpu.ids\|Warning\|_warn" | grep -A1 "rows  8192\|rows  5376"; done
