#!/bin/bash
# round 4, GPU call 1: the restructured sharded step on hardware (tests, capture stress, proxy, default line)
set -u
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
TAG=${1:-r04a}
O=gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
python -c "import torch;print(torch.__version__, torch.cuda.get_device_name(0))" > $O/box.txt 2>&1
( timeout 900 python -m pytest tests/test_sharded_gpu.py -m gpu -x -q 2>&1 | tail -15 ) > $O/test_sharded_gpu.log; echo "sharded tests: $(tail -1 $O/test_sharded_gpu.log)"
( timeout 600 python scripts/capture_stress.py 40 2>&1 | tail -5 ) > $O/capture_stress.txt; cat $O/capture_stress.txt | tail -3
for i in 1 2; do
  TZR_BENCH_TEARDOWN=destroy timeout 400 python bench.py --force-sharded --replicate-small --global-batch 8192 --steps 50 --warmup 12 --no-cpu-baseline --n1-ms 0.5955 --projection-world 8 > $O/proxy8192_$i.json 2> $O/proxy8192_$i.err; echo "proxy $i rc=$?"; cut -c1-400 $O/proxy8192_$i.json
done
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"; cut -c1-600 $O/bench_default.json
