#!/bin/bash
# gpurun -- bash scripts/r06/gpu_side.sh <tag>: the dense tail of the backward on a second stream: the equivalence test, the default step
# with it and without (--no-side-stream), same box, two rounds; the step's trace with it
tag=${1:-r06ag}; out=gpurun_out/$tag; mkdir -p $out
timeout 900 python -m pytest tests/test_fused_adam.py tests/test_graph_pipeline_gpu.py -x -q -m gpu > $out/gpu_tests.log 2>&1; echo "tests rc=$?"; tail -3 $out/gpu_tests.log
for rep in 1 2; do
for f in "" "--no-side-stream"; do
  n=$( [ -z "$f" ] && echo side || echo one )
  timeout 600 python bench.py --steps 200 --warmup 10 --no-secondary --no-cpu-baseline --no-e2e $f > $out/bench_${n}_$rep.json 2> $out/bench_${n}_$rep.err; echo "$n rc=$?"
  python -c "
import json
d=json.loads(open('$out/bench_${n}_$rep.json').read().strip().splitlines()[-1]); print('$n', d['ms_per_step'], d['value'])"
done; done
timeout 600 python bench.py --global-batch 8192 --steps 400 --warmup 10 --no-secondary --no-cpu-baseline --no-e2e > $out/bench8192_side.json 2>/dev/null
timeout 600 python bench.py --global-batch 8192 --steps 400 --warmup 10 --no-secondary --no-cpu-baseline --no-e2e --no-side-stream > $out/bench8192_one.json 2>/dev/null
python -c "
import json
for n in ('side','one'):
    d=json.loads(open('$out/bench8192_%s.json' % n).read().strip().splitlines()[-1]); print('8192', n, d['ms_per_step'])"
