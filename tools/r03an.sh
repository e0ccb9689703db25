#!/bin/bash
mkdir -p gpurun_out
for c in cfg3 cfg4; do timeout 600 python tools/train_kernels.py $c 2>&1 | grep -v "amdgpu.ids\|Warning\|_warn_once" | tee gpurun_out/r03an_train_kernels_$c.log | cut -c1-175; done
