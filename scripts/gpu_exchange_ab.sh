#!/bin/bash
# Sharded step on a 1-rank RCCL group, one box: exact exchange vs capacity-bounded vs capacity + whole-step hipGraph,
# at global batch 65536 and 8192; the new GPU tests; the side-stream plan stress with one workgroup per heavy bucket;
# the delta tracker's per-step cost.
set -u
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
TAG=${1:-r02p}
R=$PWD; O=$R/gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_sharded_gpu.py tests/test_index_parity.py tests/test_plan_invariants.py tests/test_graph_pipeline_gpu.py -m gpu -q -x --durations=8 > $O/gpu_tests_new.log 2>&1; tail -3 $O/gpu_tests_new.log
timeout 400 python -m pytest tests/test_pooled_parity.py -m gpu -q -k "plan" > $O/gpu_tests_plan.log 2>&1; tail -2 $O/gpu_tests_plan.log
for B in 65536 8192; do
  COMMON="--force-sharded --replicate-small --no-cpu-baseline --global-batch $B --secondary-global-batch 0"
  timeout 200 python bench.py $COMMON 2>> $O/bench.err | tail -1 > $O/sharded_w1_exact_b$B.json; echo "exact $B rc=$?"
  timeout 200 python bench.py $COMMON --exchange capacity 2>> $O/bench.err | tail -1 > $O/sharded_w1_capacity_b$B.json; echo "capacity $B rc=$?"
  timeout 200 python bench.py $COMMON --exchange capacity --step-graph 2>> $O/bench.err | tail -1 > $O/sharded_w1_stepgraph_b$B.json; echo "stepgraph $B rc=$?"
done
for f in $O/sharded_w1_*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); print(sys.argv[1].split('/')[-1], round(d["ms_per_step"],4), "ms", d.get("exchange"), d.get("launch"))
except Exception as e:
    print(sys.argv[1], "unreadable", e)
PY
done
TZR_ONE_WG_HEAVY=1 timeout 240 python scripts/zipf_debug.py 100 0:0 > $O/zipf_side_stream_one_wg.log 2>&1; tail -1 $O/zipf_side_stream_one_wg.log
timeout 200 python scripts/bench_delta.py > $O/bench_delta.log 2>&1; tail -2 $O/bench_delta.log
tail -5 $O/bench.err
