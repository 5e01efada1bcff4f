#!/bin/bash
mkdir -p gpurun_out/r04ab
PROF_KINDS=wgrad timeout 120 python scripts/bench_interaction_top.py --prof > gpurun_out/r04ab/phase_clocks_wgrad.txt 2>&1
tail -30 gpurun_out/r04ab/phase_clocks_wgrad.txt
