#!/bin/bash
mkdir -p gpurun_out/r04z
WG_DEBUG=8,16,23,17,22 timeout 120 python scripts/bench_interaction_top.py 65536 > gpurun_out/r04z/bench_wgrad_phases.txt 2>&1
grep "wgrad" gpurun_out/r04z/bench_wgrad_phases.txt
