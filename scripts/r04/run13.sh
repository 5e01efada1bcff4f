#!/bin/bash
mkdir -p gpurun_out/r04w
timeout 600 python -m pytest tests/test_interaction_top.py -x -q -m gpu > gpurun_out/r04w/test_interaction_top.txt 2>&1
tail -3 gpurun_out/r04w/test_interaction_top.txt
timeout 300 python scripts/bench_interaction_top.py 65536,8192 > gpurun_out/r04w/bench_interaction_top.txt 2>&1
cat gpurun_out/r04w/bench_interaction_top.txt | tail -8
