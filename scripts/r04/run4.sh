#!/bin/bash
# round 4: timing of tzr_pooled_bwd_direct variants at 8192 / 16384 per rank (bench.py embedding stages + step)
set -u
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
TAG=${1:-r04g}
O=gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
( timeout 1200 python -m pytest tests/test_pooled_parity.py -m gpu -x -q -k "backward or fp16 or rowwise" 2>&1 | tail -4 ) > $O/test_pooled_parity.log; echo "pooled parity: $(tail -1 $O/test_pooled_parity.log)"
for B in 8192 16384; do
 for cfg in "bwd_direct=-1" "bwd_direct=1" "bwd_direct=1,bwd_direct_waves=3" "bwd_direct=1,bwd_direct_ch=320,bwd_direct_waves=3" "bwd_direct=1,bwd_direct_ch=384,bwd_direct_waves=3" "bwd_direct=1,bwd_direct_ch=192" "bwd_direct=1,bwd_direct_ch=512"; do
   TZR_TUNE=$cfg timeout 300 python bench.py --global-batch $B --steps 40 --warmup 5 --no-e2e --no-cpu-baseline --no-secondary 2> $O/err.txt | tail -1 > $O/out.json
   python - <<PY
import json
try:
    d=json.load(open("$O/out.json"))
    e=d["embedding"]; print("B=$B $cfg: step %.4f ms  fwd %.1f plan %.1f apply %.1f us  frac %.3f" % (d["ms_per_step"], 1e3*e["fwd_ms"], 1e3*e["bwd_plan_ms"], 1e3*e["bwd_apply_ms"], e["frac_of_8TBps"]))
except Exception as ex:
    print("B=$B $cfg: FAILED", ex); print(open("$O/err.txt").read()[-1500:])
PY
 done
done
