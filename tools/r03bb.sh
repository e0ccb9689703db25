#!/bin/bash
mkdir -p gpurun_out
echo "== module tests"; timeout 900 python -m pytest tests/test_modules_gpu.py -q -x 2>&1 | tail -2 | cut -c1-220
timeout 600 python tools/train_kernels.py cfg4 2>&1 | grep -v "amdgpu.ids\|Warning\|_warn_once" | tee gpurun_out/r03bb_train_kernels_cfg4.log | head -16 | cut -c1-150
timeout 900 python tools/module_bench.py cfg4 > gpurun_out/r03bb_module_bench_cfg4.jsonl 2>gpurun_out/r03bb_module_bench.err; python - <<'PY'
import json
for l in open("gpurun_out/r03bb_module_bench_cfg4.jsonl"):
    r = json.loads(l)
    print(r["config"], r["what"].split("bf16, ")[-1][:80], "| ms", r["ms"], r["kernel_us"], "launches", r["launches"])
PY
