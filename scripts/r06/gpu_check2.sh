#!/bin/bash
# whole -m gpu suite + the default line and its secondaries with the one-launch index plan as the default
set -u
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
TAG=${1:-r06i}
O=$PWD/gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --durations=5 > $O/gpu_tests.log 2>&1; tail -3 $O/gpu_tests.log
timeout 900 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python - <<PY
import json; d=json.load(open("$O/bench.json")); print(d["ms_per_step"], d["value"]); r=d["roofline"]; print(r["frac"], r["launch_ms"], [ (k["stage"][:20], round(k["launch_ms"]*1e3,1)) for k in r["kernels"]])
s=d.get("secondary") or {}
for k,v in s.items():
    if isinstance(v, dict): print(k, {kk: (round(vv,4) if isinstance(vv,float) else vv) for kk,vv in v.items() if kk in ("ms_per_step","graph_ms_per_step","value")}, (v.get("embedding") or {}).get("frac_of_8TBps"))
PY
TZR_BWD_PLAN=exact timeout 600 python bench.py --no-cpu-baseline --no-secondary --no-e2e > $O/bench_exact_plan.json 2>> $O/bench.err
python - <<PY
import json; d=json.load(open("$O/bench_exact_plan.json")); print("exact plan:", d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["launch_ms"])
PY
