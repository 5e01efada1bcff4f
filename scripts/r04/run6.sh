#!/bin/bash
# round 4: full GPU suite + default line + sharded proxy with the one-launch backward
set -u
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
TAG=${1:-r04l}
O=gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
( timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 ) > $O/gpu_tests.log; echo "gpu tests: $(tail -1 $O/gpu_tests.log)"
for i in 1 2; do
  timeout 400 python bench.py --force-sharded --replicate-small --global-batch 8192 --steps 50 --warmup 12 --no-cpu-baseline --n1-ms 0.5955 --projection-world 8 2> $O/proxy8192_$i.err | tail -1 > $O/proxy8192_$i.json; echo "proxy $i rc=$?"; cut -c1-330 $O/proxy8192_$i.json
done
timeout 900 python bench.py --steps 20 --warmup 5 2> $O/bench_default.err | tail -1 > $O/bench_default.json; echo "bench rc=$?"; cut -c1-400 $O/bench_default.json
