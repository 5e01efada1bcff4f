#!/bin/bash
# phase attribution of tzr_bwd_direct_kernel at B = 8192 (stop behind geometry / id walk / sort)
set -u
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
TAG=${1:-r04i}
O=gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
for B in 8192 16384; do
 for cfg in "bwd_direct=-1" "bwd_direct=1" "bwd_direct=1,bwd_direct_debug=3" "bwd_direct=1,bwd_direct_ch=192" "bwd_direct=1,bwd_direct_ch=128"; do
   TZR_TUNE=$cfg timeout 300 python bench.py --global-batch $B --steps 10 --warmup 3 --no-e2e --no-cpu-baseline --no-secondary --no-graph 2> $O/err.txt | tail -1 > $O/out.json
   python - <<PY
import json
try:
    d=json.load(open("$O/out.json"))
    e=d["embedding"]; print("B=$B $cfg: fwd %.1f plan %.1f apply %.1f us" % (1e3*e["fwd_ms"], 1e3*e["bwd_plan_ms"], 1e3*e["bwd_apply_ms"]))
except Exception as ex:
    print("B=$B $cfg: FAILED", ex); print(open("$O/err.txt").read()[-1500:])
PY
 done
done
