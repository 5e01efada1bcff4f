#!/bin/bash
# why does bench_interaction.py (autograd) time the n = 40 backward at 2x the C-ABI loop?  kernel trace of both; n = 9 / D = 128 grids
mkdir -p gpurun_out/r03bt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/r03bt/trace_bi -- python /root/repo/scripts/bench_interaction.py > /root/repo/gpurun_out/r03bt/bi_traced.txt 2>&1
cd /root/repo
f=$(find gpurun_out/r03bt/trace_bi -name "*kernel_stats.csv" | head -1); head -12 "$f"
IA_GEN_SHAPES=8x128,39x16 timeout 300 python scripts/bench_interaction_gen.py 0 1024 2048 4096 8192 > gpurun_out/r03bt/bench_interaction_gen5.txt 2>&1
cat gpurun_out/r03bt/bench_interaction_gen5.txt
find gpurun_out/r03bt/trace_bi -name "*.csv" ! -name "*stats*" -delete
