#!/bin/bash
# rocprofv3 kernel statistics of the module-level steps (decode of the 8 LLM layers; sampling step of the 13 blocks)
mkdir -p gpurun_out
root=$PWD
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/prof_r03by_decode -- python $root/tools/decode_kernels.py 1 > $root/gpurun_out/prof_r03by_decode.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/prof_r03by_sample -- python $root/tools/sample_kernels.py > $root/gpurun_out/prof_r03by_sample.log 2>&1
cd $root
for t in decode sample; do f=$(find gpurun_out/prof_r03by_$t -name "*kernel_stats.csv" | head -1); echo "== $t $f"; head -12 "$f" | cut -c1-170; done
