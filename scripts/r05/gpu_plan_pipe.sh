#!/bin/bash
# the default step with the backward's index plan as a graph of its own, replayed on a second stream during the previous step,
# against the plan inside the step's graph (--serial-plan); same box, alternating
set -u
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
TAG=${1:-r05aj}; R=$PWD; O=$R/gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
for rep in 1 2 3; do for mode in "--serial-plan" ""; do
  n=$(echo "x$mode" | tr -d ' -')
  timeout 300 python bench.py --steps 200 --warmup 5 --no-cpu-baseline --no-e2e --no-secondary $mode > $O/step_$n.$rep.json 2> $O/step_$n.$rep.err
  python - "$O/step_$n.$rep.json" "$mode" <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    print(sys.argv[2] or "plan ahead ", "ms_per_step", round(d["ms_per_step"],4), "loss", d.get("final_loss"), d["launch"][:40])
except Exception as e:
    print("parse failed", e); print(open(sys.argv[1].replace(".json",".err")).read()[-1500:])
PY
done; done
