#!/bin/bash
for g in 512 256 128 1024 2048; do echo "== bwd grid $g"; MMFS_NORM_BWD_GRID=$g timeout 120 python tools/norm_bench.py 2>&1 | grep rows; done
