#!/bin/bash
O=gpurun_out/${1:-r04an}; mkdir -p $O
PROF_KINDS=${2:-bwd,fwd} timeout 120 python scripts/bench_interaction_top.py --prof > $O/phase_clocks.txt 2>&1
grep -A18 "^bwd:\|^fwd:" $O/phase_clocks.txt | grep "shader\|all\|wave  0\|wave  4\|wave  8\|wave 12"
