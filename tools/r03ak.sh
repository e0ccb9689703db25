#!/bin/bash
mkdir -p gpurun_out
timeout 300 python tools/decode_kernels.py 1 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03ak_decode_kernels.log
