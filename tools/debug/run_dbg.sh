for r in 1 2 3 4 5 6; do python tools/debug/q8_scan.py 2>&1 | grep -v amdgpu.ids | cut -c1-120 | tail -3; done
python -m pytest tests/test_op_gpu.py -m gpu -q -x -k sliced 2>&1 | tail -3
for w in cfg2_sd_real cfg3_llm_n1 cfg5_llm_n4 cfg2_northstar enc_injector; do MMFS_FWD_ALGO=q8 python bench.py --no-cpu-baseline --steps 30 --warmup 10 --workload $w 2>/dev/null | python -c "
import sys, json
r = json.loads(sys.stdin.read()); print('$w', r['ms_per_step'], r['kernels_mean_us'].get('msda_fwd'))"; done
