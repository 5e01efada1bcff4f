#!/bin/bash
O=gpurun_out/${1:-r04af}; mkdir -p $O
timeout 120 python -m pytest tests/test_interaction_top.py -x -q -m gpu -k wgrad > $O/test_wgrad.txt 2>&1
tail -2 $O/test_wgrad.txt
WG_DEBUG=${2:-1,6,7,16} timeout 120 python scripts/bench_interaction_top.py 65536,8192 > $O/bench_wgrad_phases.txt 2>&1
grep "wgrad" $O/bench_wgrad_phases.txt
PROF_KINDS=wgrad timeout 120 python scripts/bench_interaction_top.py --prof > $O/phase_clocks_wgrad.txt 2>&1
grep -A5 "wgrad:" $O/phase_clocks_wgrad.txt; tail -1 $O/phase_clocks_wgrad.txt
