#!/bin/bash
mkdir -p gpurun_out
echo "== module tests"; timeout 900 python -m pytest tests/test_modules_gpu.py tests/test_zz_graph_gpu.py -q -x 2>&1 | tail -4 | cut -c1-220
timeout 300 python tools/decode_kernels.py 1 2>&1 | grep -v "amdgpu.ids\|Warning\|_warn_once" | tee gpurun_out/r03br_decode_kernels.log | head -8 | cut -c1-150
timeout 900 python tools/module_bench.py cfg3 cfg4 > gpurun_out/r03br_module_bench_cfg3_cfg4.jsonl 2>gpurun_out/r03br_module_bench.err; python - <<'PY'
import json
for l in open("gpurun_out/r03br_module_bench_cfg3_cfg4.jsonl"):
    r = json.loads(l)
    if "forward+backward" in r["what"] and "2048" not in r["what"]: continue
    print(r["config"], r["what"].split("B=")[1][:12], r["what"].split("bf16, ")[-1][:80], "| ms", r["ms"], r["kernel_us"], "launches", r["launches"])
PY
