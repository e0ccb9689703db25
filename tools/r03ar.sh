#!/bin/bash
mkdir -p gpurun_out
echo "== module tests"; timeout 900 python -m pytest tests/test_modules_gpu.py -q -x 2>&1 | tail -2
timeout 900 python tools/module_bench.py cfg3 cfg4 > gpurun_out/r03ar_module_bench_cfg3_cfg4.jsonl 2>gpurun_out/r03ar_module_bench.err; python - <<'PY'
import json
for l in open("gpurun_out/r03ar_module_bench_cfg3_cfg4.jsonl"):
    r = json.loads(l); print(r["config"], r["what"].split("bf16, ")[-1][:90], "| ms", r["ms"], r["kernel_us"], "launches", r["launches"])
PY
