#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
TAG=${1:-r03m}; R=$PWD; O=$R/gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_graph_pipeline_gpu.py tests/test_pooled_parity.py tests/test_plan_invariants.py tests/test_sharded_gpu.py -m gpu -q -x > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log
timeout 600 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -3 $O/bench.err
python - <<PY
import json
d=json.load(open("$O/bench.json"))
print("value", d["value"], "ms", d["ms_per_step"], "frac", d["roofline"]["frac"], "traffic", d["roofline"]["traffic"], d["roofline"]["traffic_source"][:90])
print(json.dumps(d["secondary"], indent=1)[:2500])
PY
