#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
TAG=${1:-r03bm}; R=$PWD; O=$R/gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_sharded_gpu.py tests/test_zz_mixed_sharded_world1.py tests/test_graph_pipeline_gpu.py -m gpu -q -x > $O/tests.log 2>&1; echo "tests rc=$?"; grep -E "passed|failed" $O/tests.log | tail -1
for gb in 65536 8192; do
timeout 300 python bench.py --force-sharded --replicate-small --no-cpu-baseline --global-batch $gb 2>> $O/bench.err | tail -1 > $O/sharded_w1_proxy_bench_$gb.json; python -c "
import json; d=json.load(open('$O/sharded_w1_proxy_bench_$gb.json')); print($gb, round(d['value']/1e6,2), round(d['ms_per_step'],4))"
done
timeout 400 python bench.py --no-cpu-baseline --no-secondary > $O/bench.json 2>> $O/bench.err; python -c "
import json; d=json.load(open('$O/bench.json')); print(round(d['value']/1e6,2), d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic'])"
