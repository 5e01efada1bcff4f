#!/bin/bash
# generalised interaction backward: next block's operand loads ahead of the stores
mkdir -p gpurun_out/r03bt
IA_GEN_SHAPES=39x16,63x32,26x64,16x64,8x128,15x64,11x32 timeout 300 python scripts/bench_interaction_gen.py 0 1024 4096 > gpurun_out/r03bt/bench_interaction_gen7.txt 2>&1
cat gpurun_out/r03bt/bench_interaction_gen7.txt
