#!/bin/bash
# The default step with the backward index plan (ids only) as a parallel branch of the step's hipGraph, against the serial form
set -u
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
TAG=${1:-r05w}; R=$PWD; O=$R/gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
for rep in 1 2; do for mode in "" "--async-plan"; do
  timeout 300 python bench.py --steps 200 --warmup 5 --no-cpu-baseline --no-e2e --no-secondary $mode > $O/step$mode.$rep.json 2> $O/step$mode.$rep.err
  python - "$O/step$mode.$rep.json" <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    print(sys.argv[1].split("/")[-1], "ms_per_step", round(d["ms_per_step"],4), "loss", d.get("final_loss"))
except Exception as e:
    print("parse failed", e); print(open(sys.argv[1].replace(".json",".err")).read()[-1500:])
PY
done; done
