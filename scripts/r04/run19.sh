#!/bin/bash
mkdir -p gpurun_out/r04ac
timeout 120 python -m pytest tests/test_interaction_top.py -x -q -m gpu -k wgrad > gpurun_out/r04ac/test_wgrad.txt 2>&1
tail -2 gpurun_out/r04ac/test_wgrad.txt
WG_DEBUG=1,6,7,16 timeout 120 python scripts/bench_interaction_top.py 65536,8192 > gpurun_out/r04ac/bench_wgrad_phases.txt 2>&1
grep "wgrad" gpurun_out/r04ac/bench_wgrad_phases.txt
PROF_KINDS=wgrad timeout 120 python scripts/bench_interaction_top.py --prof > gpurun_out/r04ac/phase_clocks_wgrad.txt 2>&1
grep -A5 "wgrad:" gpurun_out/r04ac/phase_clocks_wgrad.txt; tail -1 gpurun_out/r04ac/phase_clocks_wgrad.txt
