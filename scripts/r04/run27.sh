#!/bin/bash
O=gpurun_out/${1:-r04am}; mkdir -p $O
timeout 200 python -m pytest tests/test_interaction_top.py -x -q -m gpu > $O/test_interaction_top.txt 2>&1; tail -2 $O/test_interaction_top.txt
timeout 120 python scripts/bench_interaction_top.py 65536,8192 > $O/bench_interaction_top.txt 2>&1; grep "fused\|wgrad" $O/bench_interaction_top.txt
