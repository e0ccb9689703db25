#!/bin/bash
mkdir -p gpurun_out
echo "== fuzz, default routes"; timeout 600 python tests/fuzz_op.py 300 31 > gpurun_out/r03aj_fuzz_default.log 2>&1; grep -c "^ok" gpurun_out/r03aj_fuzz_default.log; grep "FAIL\|cases within" gpurun_out/r03aj_fuzz_default.log | head -10
echo "== fuzz, big"; timeout 900 python tests/fuzz_op.py 60 32 big > gpurun_out/r03aj_fuzz_big.log 2>&1; grep "FAIL\|cases within" gpurun_out/r03aj_fuzz_big.log | head -10
echo "== fuzz, workgroup-local grad_value on"; MMFS_GV_ALGO=on MMFS_GV_MIN_NQ=16 timeout 600 python tests/fuzz_op.py 250 33 > gpurun_out/r03aj_fuzz_gv.log 2>&1; grep "FAIL\|cases within" gpurun_out/r03aj_fuzz_gv.log | head -10
echo "== fuzz, big, gv on"; MMFS_GV_ALGO=on timeout 900 python tests/fuzz_op.py 40 34 big > gpurun_out/r03aj_fuzz_gv_big.log 2>&1; grep "FAIL\|cases within" gpurun_out/r03aj_fuzz_gv_big.log | head -10
