#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
TAG=${1:-r05k}; R=$PWD; O=$R/gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_sharded_gpu.py -m gpu -x -q -k "whole_step" > $O/gpu_tests_native.log 2>&1; echo "tests rc=$?"; tail -25 $O/gpu_tests_native.log | cut -c1-300
for mode in "" "--no-native-driver"; do
  timeout 400 python bench.py --gpus 1 --force-sharded --replicate-small --global-batch 8192 --steps 60 --warmup 12 --no-cpu-baseline --no-e2e --projection-world 8 $mode > $O/proxy8192$mode.json 2> $O/proxy8192$mode.err
  python - "$O/proxy8192$mode.json" <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    print(sys.argv[1].split("/")[-1], "ms_per_step", round(d["ms_per_step"],4), "host_queue", round(d["host_queue_ms_per_step"],4), d.get("exchange"), d.get("launch","")[:80])
except Exception as e:
    print("parse failed", e); print(open(sys.argv[1].replace(".json",".err")).read()[-1500:])
PY
done
