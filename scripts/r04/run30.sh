#!/bin/bash
O=gpurun_out/${1:-r04ap}; mkdir -p $O
IT_FWD_STAGGER=0,1,2 IT_STAGGER=0,1,2 timeout 120 python scripts/bench_interaction_top.py 65536 > $O/bench_interaction_top_stagger.txt 2>&1; grep "stagger\|fused" $O/bench_interaction_top_stagger.txt
