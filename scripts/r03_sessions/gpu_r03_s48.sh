#!/bin/bash
# whole-step path with the overlapped order forced on one rank: bit-identical to the exact step? what do the two extra graphs cost?
mkdir -p gpurun_out/r03bx
timeout 600 python -m pytest tests/test_sharded_gpu.py -q -m gpu -k "whole_step_graph" > gpurun_out/r03bx/pytest_whole_step.txt 2>&1; grep -E "passed|failed" gpurun_out/r03bx/pytest_whole_step.txt | tail -1
for o in off on; do timeout 300 python bench.py --force-sharded --replicate-small --no-cpu-baseline --global-batch 8192 --overlap-collectives $o 2>/dev/null | tail -1 > gpurun_out/r03bx/sharded_8192_overlap_$o.json; python -c "import json; j=json.loads(open('gpurun_out/r03bx/sharded_8192_overlap_$o.json').read()); print('$o', j['ms_per_step'], j['launch'][:80])"; done
