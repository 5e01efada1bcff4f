#!/bin/bash
mkdir -p gpurun_out/r04y
timeout 120 python -m pytest tests/test_interaction_top.py -x -q -m gpu -k wgrad > gpurun_out/r04y/test_wgrad.txt 2>&1
tail -2 gpurun_out/r04y/test_wgrad.txt
WG_DEBUG=1,2,4,6,7,3 timeout 120 python scripts/bench_interaction_top.py 65536,8192 > gpurun_out/r04y/bench_wgrad_phases.txt 2>&1
grep "wgrad" gpurun_out/r04y/bench_wgrad_phases.txt
