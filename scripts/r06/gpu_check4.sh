#!/bin/bash
# gpurun -- bash scripts/r06/gpu_check4.sh <tag>: after host-side changes: fused-Adam / sharded / pipeline tests on the GPU, smoke(), the default line
tag=${1:-r06ae}; out=gpurun_out/$tag; mkdir -p $out
timeout 1200 python -m pytest tests/test_fused_adam.py tests/test_sharded_gpu.py tests/test_graph_pipeline_gpu.py tests/test_dense_glue.py -x -q -m gpu > $out/gpu_tests.log 2>&1; echo "tests rc=$?"; tail -2 $out/gpu_tests.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $out/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $out/smoke.log
timeout 900 python bench.py --no-secondary > $out/bench.json 2> $out/bench.err; echo "bench rc=$?"
python -c "
import json
d=json.loads(open('$out/bench.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], d['roofline']['frac'], d['roofline']['traffic'])"
