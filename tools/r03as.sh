#!/bin/bash
mkdir -p gpurun_out
timeout 300 python tools/sample_kernels.py 2>&1 | grep -v "amdgpu.ids\|Warning\|_warn_once" | tee gpurun_out/r03as_sample_kernels.log | cut -c1-185
