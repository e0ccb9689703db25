#!/bin/bash
mkdir -p gpurun_out
for t in 256 512; do
  echo "== target $t"; MMFS_GV_TARGET_WGS=$t MMFS_MSDA_LIB=$PWD/mm-interleaved_amd/csrc/build/exp/gprof.so timeout 300 python tools/gv_prof.py cfg2_northstar 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r03n_gv_phase_clocks.log
done
echo "== llm"; MMFS_MSDA_LIB=$PWD/mm-interleaved_amd/csrc/build/exp/gprof.so timeout 300 python tools/gv_prof.py cfg5_llm_n4 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r03n_gv_phase_clocks.log
