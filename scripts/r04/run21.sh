#!/bin/bash
mkdir -p gpurun_out/r04ae
WG_DEBUG=64 timeout 120 python scripts/bench_interaction_top.py 65536 > gpurun_out/r04ae/bench_wgrad_phases.txt 2>&1
grep "wgrad" gpurun_out/r04ae/bench_wgrad_phases.txt
PROF_KINDS=wgrad timeout 120 python scripts/bench_interaction_top.py --prof > gpurun_out/r04ae/phase_clocks_wgrad.txt 2>&1
grep -A5 "wgrad:" gpurun_out/r04ae/phase_clocks_wgrad.txt; tail -1 gpurun_out/r04ae/phase_clocks_wgrad.txt
