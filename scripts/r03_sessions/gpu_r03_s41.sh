#!/bin/bash
# 17-32-row (D != 16) interaction backward: the same variants
mkdir -p gpurun_out/r03bt
IA_GEN_SMALL=1 IA_GEN_VARS=0,1,3,7 timeout 300 python scripts/bench_interaction_gen.py 0 1024 1792 2048 > gpurun_out/r03bt/bench_interaction_gen4.txt 2>&1
tail -70 gpurun_out/r03bt/bench_interaction_gen4.txt
