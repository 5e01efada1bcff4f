#!/bin/bash
# 33-64-row interaction backward: var 7 at two workgroups per CU (4-wave form), pair-gradient-only prefetch (var 11)
mkdir -p gpurun_out/r03bt
IA_GEN_VARS=3,7,11 timeout 300 python scripts/bench_interaction_gen.py 512 768 1024 > gpurun_out/r03bt/bench_interaction_gen3.txt 2>&1
tail -40 gpurun_out/r03bt/bench_interaction_gen3.txt
