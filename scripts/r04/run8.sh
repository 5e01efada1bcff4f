#!/bin/bash
# round 4: sharded proxy after the root_loss fix; host queueing time; graph_input_dist A/B
set -u
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
TAG=${1:-r04o}
O=gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
for i in 1 2; do
  timeout 400 python bench.py --force-sharded --replicate-small --global-batch 8192 --steps 60 --warmup 12 --no-cpu-baseline --n1-ms 0.5846 --projection-world 8 2> $O/proxy.err | tail -1 > $O/proxy8192_$i.json
  python - <<PY
import json
d=json.load(open("$O/proxy8192_$i.json")); print("proxy run $i: %.4f ms/step, host queue %.4f ms/step" % (d["ms_per_step"], d["host_queue_ms_per_step"]), d["projection"]["scaling_vs_n1"])
PY
done
timeout 400 python bench.py --global-batch 8192 --steps 60 --warmup 5 --no-cpu-baseline --no-e2e --no-secondary 2> $O/c2.err | tail -1 > $O/config2.json
python - <<PY
import json
d=json.load(open("$O/config2.json")); print("config2: %.4f ms/step host %.4f" % (d["ms_per_step"], d["host_queue_ms_per_step"]), d["embedding"])
PY
