#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out; export TMPDIR=/tmp
for mode in "" "--no-graph"; do
  for rep in 1 2; do
    timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline $mode > gpurun_out/b.json 2> gpurun_out/b.err; rc=$?
    python - <<PY
import json
try:
    d=json.load(open("gpurun_out/b.json")); e=d.get("embedding",{})
    print("mode=[$mode] rep$rep rc=$rc", round(d["value"]/1e6,2), "M/s", round(d["ms_per_step"],4), "ms", d["launch"], "fwd/plan/apply", round(e.get("fwd_ms",0)*1e3), round(e.get("bwd_plan_ms",0)*1e3), round(e.get("bwd_apply_ms",0)*1e3), "loss", d["final_loss"])
except Exception as ex:
    print("mode=[$mode] rc=$rc FAILED", ex); print(open("gpurun_out/b.err").read()[-1500:])
PY
  done
done
