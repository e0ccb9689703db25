#!/bin/bash
# final check of the round: whole GPU suite, smoke, the driver's command twice
mkdir -p gpurun_out
echo "== all gpu tests"; timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r03bq_pytest_all.log 2>&1; tail -2 gpurun_out/r03bq_pytest_all.log | cut -c1-200
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
for i in 1 2; do
timeout 250 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_r03bq_driver_$i.json 2>/dev/null
python - gpurun_out/bench_r03bq_driver_$i.json <<'PY'
import json, sys
r = json.load(open(sys.argv[1])); print(r["ms_per_step"], r["value"], r["roofline"]["kernel"], r["roofline"]["frac"], r["kernels_frac"], r["cpu_baseline"]["value"] if "cpu_baseline" in r else None)
PY
done
