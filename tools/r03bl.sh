#!/bin/bash
# randomised parity sweep of the second closing build (persistent workgroups, sort head groups) + the new persistent test
mkdir -p gpurun_out
echo "== persistent test"; timeout 600 python -m pytest tests/test_op_gpu.py -q -x -k "persistent or lds_levels or taps_mma" 2>&1 | tail -2 | cut -c1-200
echo "== fuzz, 8 persistent workgroups, LDS-resident kernels wherever they apply"; MMFS_MMA_GRID=8 MMFS_FWD_ALGO=mma MMFS_TAPS_ALGO=mma timeout 600 python tests/fuzz_op.py 150 35 > gpurun_out/r03bl_fuzz_grid8.log 2>&1; grep "FAIL\|cases within" gpurun_out/r03bl_fuzz_grid8.log | head -10
echo "== fuzz, default routes"; timeout 600 python tests/fuzz_op.py 300 31 > gpurun_out/r03bl_fuzz_default.log 2>&1; grep -c "^ok" gpurun_out/r03bl_fuzz_default.log; grep "FAIL\|cases within" gpurun_out/r03bl_fuzz_default.log | head -10
echo "== fuzz, big"; timeout 900 python tests/fuzz_op.py 60 32 big > gpurun_out/r03bl_fuzz_big.log 2>&1; grep "FAIL\|cases within" gpurun_out/r03bl_fuzz_big.log | head -10

