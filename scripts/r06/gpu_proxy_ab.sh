#!/bin/bash
# gpurun -- bash scripts/r06/gpu_proxy_ab.sh <tag>: the 8192-per-rank sharded step (1-rank RCCL proxy) with the dense gradients packed by
# the one fused launch from partial sums (default) and with the finishing launches + concatenation (--no-fuse-finish), same box;
# the world-1 RCCL tests
tag=${1:-r06aa}; out=gpurun_out/$tag; mkdir -p $out
timeout 900 python -m pytest tests/test_sharded_gpu.py tests/test_fused_adam.py -x -q -m gpu > $out/gpu_tests.log 2>&1; echo "tests rc=$?"; tail -3 $out/gpu_tests.log
for rep in 1 2; do
for f in "" "--no-fuse-finish"; do
  n=$( [ -z "$f" ] && echo fused || echo finish )
  timeout 600 python bench.py --gpus 1 --force-sharded --replicate-small --global-batch 8192 --steps 200 --warmup 12 --no-cpu-baseline --no-e2e --projection-world 8 $f > $out/proxy_${n}_$rep.json 2> $out/proxy_${n}_$rep.err; echo "$n rc=$?"
  python -c "
import json,sys
d=json.loads(open('$out/proxy_${n}_$rep.json').read().strip().splitlines()[-1]); print('$n', d['ms_per_step'], d.get('host_queue_ms_per_step'))"
done; done
timeout 600 python bench.py --gpus 1 --force-sharded --replicate-small --global-batch 65536 --steps 60 --warmup 12 --no-cpu-baseline --no-e2e > $out/proxy65536_fused.json 2>/dev/null; python -c "
import json
d=json.loads(open('$out/proxy65536_fused.json').read().strip().splitlines()[-1]); print('65536 fused', d['ms_per_step'])"
