#!/bin/bash
# (a) the default step kernel by kernel (timeline of the eager launch order); (b) the 65 536-per-rank sharded step on the 1-rank RCCL
# group: exact exchange (pipelined) against the capacity exchange + native step driver (re-measure the auto threshold)
set -u
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
TAG=${1:-r05v}; R=$PWD; O=$R/gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
cd /tmp
timeout 300 rocprofv3 --kernel-trace -d $O/trace -o t -- python $R/bench.py --steps 6 --warmup 4 --no-cpu-baseline --no-e2e --no-secondary --no-graph > $O/trace.log 2>&1; echo "trace rc=$?"
cd $R
DB=$(find $O/trace -name '*.db' | head -1)
python scripts/rocpd_timeline.py "$DB" 400 $O/timeline_default_step.txt
rm -rf $O/trace
for mode in "--exchange exact" "--exchange capacity --step-graph" "--exchange capacity --step-graph --no-native-driver"; do
  n=$(echo $mode | tr -d ' -')
  timeout 500 python bench.py --gpus 1 --force-sharded --replicate-small --global-batch 65536 --steps 60 --warmup 12 --no-cpu-baseline --no-e2e --projection-world 8 $mode > $O/proxy65536_$n.json 2> $O/proxy65536_$n.err
  python - "$O/proxy65536_$n.json" <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    print(sys.argv[1].split("/")[-1], "ms_per_step", round(d["ms_per_step"],4), "host_queue", round(d["host_queue_ms_per_step"],4), d.get("exchange"), d.get("launch","")[:80])
except Exception as e:
    print("parse failed", e); print(open(sys.argv[1].replace(".json",".err")).read()[-1500:])
PY
done
