# which of msda_fwd_q8's guards matter today (r06): the stress test's "slices" cases per experimental build
#   tools/exp_build1.sh q8_builtin msda_fwd_q8 "-DQ8_BUILTIN_MFMA"   (etc.), then: bash tools/debug/q8_matrix.sh [variants...]
vs=${@:-base q8_noguard q8_builtin q8_builtin_noguard q8_read2 q8_noprio}
for v in $vs; do
  if [ $v = base ]; then lib=$PWD/mm-interleaved_amd/libmmfs_msda.so; else lib=$PWD/mm-interleaved_amd/csrc/build/exp/$v.so; fi
  echo "== $v"
  MMFS_MSDA_LIB=$lib timeout 600 python -m pytest tests/test_stress_gpu.py -q -k "slices" 2>&1 | grep -E "^E  |passed|failed" | cut -c1-260 | head -${Q8_LINES:-12}
done
