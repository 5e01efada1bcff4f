#!/bin/bash
# 33-64-row interaction backward: 32-bit offsets; cross-sample prefetch (var 7); grids
mkdir -p gpurun_out/r03bt
IA_GEN_VARS=0,3,7 timeout 300 python scripts/bench_interaction_gen.py 0 512 768 1024 > gpurun_out/r03bt/bench_interaction_gen2.txt 2>&1
timeout 200 python scripts/bench_interaction.py > gpurun_out/r03bt/bench_interaction.txt 2>&1
tail -50 gpurun_out/r03bt/bench_interaction_gen2.txt; tail -7 gpurun_out/r03bt/bench_interaction.txt
