#!/bin/bash
# 33-64-row interaction backward: variants x grids
mkdir -p gpurun_out/r03bt
IA_GEN_VARS=0,1,2,3 timeout 300 python scripts/bench_interaction_gen.py 0 768 1536 3072 > gpurun_out/r03bt/bench_interaction_gen.txt 2>&1
tail -70 gpurun_out/r03bt/bench_interaction_gen.txt
