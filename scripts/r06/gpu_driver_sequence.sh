#!/bin/bash
# gpurun -- bash scripts/r06/gpu_driver_sequence.sh <tag>: what the driver runs at round end on a fresh box: the GPU tests, smoke(), the bench line with its flags
tag=${1:-r06ay}; out=gpurun_out/$tag; mkdir -p $out
timeout 1500 python -m pytest tests/ -x -q -m gpu > $out/gpu_tests.log 2>&1; echo "tests rc=$?"; tail -1 $out/gpu_tests.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $out/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $out/smoke.log
timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_driver_flags.json 2> $out/bench.err; echo "bench rc=$?"
python -c "
import json
d=json.loads(open('$out/bench_driver_flags.json').read().strip().splitlines()[-1]); r=d['roofline']; print(d['ms_per_step'], d['value'], r['frac'], r['traffic'], d['cpu_baseline']['value'], d['steps'], d['warmup'])"
