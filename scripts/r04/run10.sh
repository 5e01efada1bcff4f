#!/bin/bash
# round 4: host-time cuts of the sharded step (raw stream handle, direct process-group calls, earlier prefetch, graphed input dist)
set -u
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
TAG=${1:-r04r}
O=gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
echo
for flags in ""; do
 for i in 1 2; do
  timeout 400 python bench.py --force-sharded --replicate-small --global-batch 8192 --steps 80 --warmup 14 --no-cpu-baseline --n1-ms 0.5651 --projection-world 8 $flags 2> $O/proxy.err | tail -1 > $O/proxy8192.json
  python - <<PY
import json
d=json.load(open("$O/proxy8192.json")); print("proxy [$flags] run $i: %.4f ms/step, host queue %.4f ms/step" % (d["ms_per_step"], d["host_queue_ms_per_step"]), d["projection"]["scaling_vs_n1"], d["exchange"])
PY
 done
done
cp $O/proxy8192.json $O/proxy8192_last.json
timeout 400 python bench.py --force-sharded --replicate-small --steps 30 --warmup 8 --no-cpu-baseline 2> $O/proxy65536.err | tail -1 > $O/proxy65536.json
python - <<PY
import json
d=json.load(open("$O/proxy65536.json")); print("proxy 65536: %.4f ms/step host %.4f" % (d["ms_per_step"], d["host_queue_ms_per_step"]))
PY
timeout 300 python scripts/r04/profile_host.py 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | head -60 > $O/host_profile.txt
