#!/bin/bash
# A/B on one box: the batch-8192 step and its embedding stages (uniform ids: no hot row anywhere) with and without the hot-row
# candidate of tzr_pooled_bwd_direct (TZR_TUNE=bwd_direct_hot=0)
set -u
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
TAG=${1:-r05ac}; R=$PWD; O=$R/gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
for rep in 1 2 3; do for hot in 1 0; do
TZR_TUNE=bwd_direct_hot=$hot timeout 300 python bench.py --global-batch 8192 --steps 200 --no-cpu-baseline --no-e2e --no-secondary > $O/bench_b8192.hot$hot.$rep.json 2>> $O/bench.err
python - $O/bench_b8192.hot$hot.$rep.json $hot <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print("hot", sys.argv[2], "ms_per_step", round(d["ms_per_step"],4), {k: round(v,4) if isinstance(v,float) else v for k,v in d["embedding"].items()})
PY
done; done
