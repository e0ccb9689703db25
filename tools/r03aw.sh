#!/bin/bash
mkdir -p gpurun_out
echo "== layout tests"; timeout 900 python -m pytest tests/test_modules_gpu.py -q -x -k "query_prep or tokens_add or layout_kernels or graphed" 2>&1 | tail -3 | cut -c1-220
for kb in 25 33; do
echo "== MMFS_QUERY_LDS_KB=$kb"
MMFS_QUERY_LDS_KB=$kb timeout 300 python tools/sample_kernels.py 2>&1 | grep -v "amdgpu.ids\|Warning\|_warn_once" | tee gpurun_out/r03aw_sample_kernels_$kb.log | grep "sampling step\|query_prep\|tokens_add" | cut -c1-110
done
