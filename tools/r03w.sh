#!/bin/bash
mkdir -p gpurun_out
run() { local name=$1; shift; timeout 300 env "$@" > gpurun_out/bench_r03w_$name.json 2> gpurun_out/bench_r03w_$name.err || echo "FAILED $name"; python - "$name" <<'PY'
import json, sys
try:
    r = json.load(open(f"gpurun_out/bench_r03w_{sys.argv[1]}.json")); print(sys.argv[1], r["ms_per_step"], r.get("kernels_mean_us"))
except Exception as e:
    print(sys.argv[1], "no result", e); print(open(f"gpurun_out/bench_r03w_{sys.argv[1]}.err").read()[-1500:])
PY
}
python -c "import torch; print(torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream,'priority_range') else 'n/a')"
run ns_base python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-events
run ns_ov_hi MMFS_BWD_OVERLAP=1 MMFS_BWD_OVERLAP_PRIORITY=-1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-events
run ns_ov_lo MMFS_BWD_OVERLAP=1 MMFS_BWD_OVERLAP_PRIORITY=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-events
run ns_ov_0 MMFS_BWD_OVERLAP=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-events
run ns_base2 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-events
