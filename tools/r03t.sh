#!/bin/bash
mkdir -p gpurun_out
MMFS_MSDA_LIB=$PWD/mm-interleaved_amd/csrc/build/exp/sprof.so timeout 300 python tools/sort_prof.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03t_sort_phase_clocks.log
