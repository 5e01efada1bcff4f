#!/bin/bash
# Sharded step on a 1-rank RCCL group, one box: exact exchange vs capacity-bounded vs capacity + whole-step hipGraphs,
# at global batch 65536 and 8192 (MODES / SKIP_TESTS select).
set -u
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
TAG=${1:-r02p}
R=$PWD; O=$R/gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
if [ -z "${SKIP_TESTS:-}" ]; then
timeout 300 python -m pytest tests/test_sharded_gpu.py -m gpu -q -x -k "whole_step or pipelined" > $O/gpu_tests_stepgraph.log 2>&1; tail -3 $O/gpu_tests_stepgraph.log
fi
for B in 65536 8192; do
  COMMON="--force-sharded --replicate-small --no-cpu-baseline --global-batch $B --secondary-global-batch 0"
  for M in ${MODES:-exact capacity stepgraph}; do
    case $M in exact) X="";; capacity) X="--exchange capacity";; stepgraph) X="--exchange capacity --step-graph";; esac
    timeout 200 python bench.py $COMMON $X 2>> $O/bench.err | tail -1 > $O/sharded_w1_${M}_b$B.json; echo "$M $B rc=$?"
  done
done
for f in $O/sharded_w1_*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); print(sys.argv[1].split('/')[-1], round(d["ms_per_step"],4), "ms", d.get("exchange"), d.get("launch"))
except Exception as e:
    print(sys.argv[1], "unreadable", e)
PY
done
grep -v "amdgpu.ids\|hostname of the client" $O/bench.err | tail -8
