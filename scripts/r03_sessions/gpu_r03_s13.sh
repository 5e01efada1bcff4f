#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
TAG=${1:-r03s}; R=$PWD; O=$R/gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_index_parity.py tests/test_sharded_gpu.py -m gpu -q -x > $O/tests.log 2>&1; echo "tests rc=$?"; grep -E "passed|failed" $O/tests.log | tail -1
timeout 300 python bench.py --force-sharded --replicate-small --no-cpu-baseline 2>> $O/bench.err | tail -1 > $O/sharded_w1_proxy_bench.json; echo "proxy rc=$?"
timeout 300 python bench.py --force-sharded --replicate-small --no-cpu-baseline --global-batch 8192 2>> $O/bench.err | tail -1 > $O/sharded_w1_proxy_bench_b8192.json; echo "proxy 8192 rc=$?"
python - <<PY
import json
for f in ("sharded_w1_proxy_bench","sharded_w1_proxy_bench_b8192"):
    d=json.load(open("$O/%s.json"%f)); print(f, round(d["ms_per_step"],4), "ms", round(d["value"]/1e6,2), "M/s", d["config"]["parallelism"][:100])
PY
