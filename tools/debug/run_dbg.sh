python -m pytest tests/test_op_gpu.py -m gpu -q -x -k sliced 2>&1 | tail -2
for w in cfg2_sd_real cfg3_llm_n1 cfg5_llm_n4 enc_injector enc_extractor; do for q in 0 512; do MMFS_FWD_Q8_QPR=$q MMFS_FWD_ALGO=q8 python bench.py --no-cpu-baseline --steps 30 --warmup 10 --workload $w 2>/dev/null | python -c "
import sys, json
r = json.loads(sys.stdin.read()); print('$w qpr=$q', r['ms_per_step'], r['kernels_mean_us'].get('msda_fwd'))"; done; done
