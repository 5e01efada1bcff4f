#!/bin/bash
# round 4, GPU call 3: the one-launch backward (tzr_pooled_bwd_direct) on hardware: parity in both forms, timing at 8192 / 16384
set -u
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
TAG=${1:-r04d}
O=gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
( timeout 1200 python -m pytest tests/test_pooled_parity.py -m gpu -x -q 2>&1 | tail -4 ) > $O/test_pooled_parity.log; echo "pooled parity: $(tail -1 $O/test_pooled_parity.log)"
for B in 8192 16384; do
 for knob in -1 1; do
  for ch in 0 128; do
   if [ $knob = -1 ] && [ $ch != 0 ]; then continue; fi
   TZR_TUNE=bwd_direct=$knob,bwd_direct_ch=$ch timeout 300 python bench.py --global-batch $B --steps 40 --warmup 5 --no-e2e --no-cpu-baseline --no-secondary 2> $O/b${B}_k${knob}_c$ch.err | tail -1 > $O/b${B}_k${knob}_c$ch.json
   python - <<PY
import json
d=json.load(open("$O/b${B}_k${knob}_c$ch.json"))
e=d["embedding"]; print("B=$B direct=$knob ch=$ch step %.4f ms  fwd %.1f plan %.1f apply %.1f us  frac %.3f" % (d["ms_per_step"], 1e3*e["fwd_ms"], 1e3*e["bwd_plan_ms"], 1e3*e["bwd_apply_ms"], e["frac_of_8TBps"]))
PY
  done
 done
done
