#!/bin/bash
# per-kernel trace stats of the default step (eager launches, 20 steps) into gpurun_out/<tag>/kernel_stats.csv + a short table
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
TAG=${1:-r05x}; R=$PWD; O=$R/gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace -o t --output-format csv -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-e2e --no-graph --no-secondary > $O/trace.log 2>&1; echo "trace rc=$?"
cd $R
S=$(find $O/trace -name '*kernel_stats.csv' | head -1); cp "$S" $O/kernel_stats.csv; rm -rf $O/trace
python - <<PY
import csv
rows=[r for r in csv.DictReader(open("$O/kernel_stats.csv")) if r["Name"].startswith("tzr_") or "tzr_" in r["Name"][:30]]
for r in sorted(rows,key=lambda r:-float(r["TotalDurationNs"]))[:22]:
    print(f"{r['Name'][:52]:52s} {r['Calls']:>5s} {float(r['AverageNs'])/1e3:8.1f}")
PY
