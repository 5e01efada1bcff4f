#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
TAG=${1:-r03v}; R=$PWD; O=$R/gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace -o t --output-format csv -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-e2e --no-graph --no-secondary > $O/trace.log 2>&1; echo "trace rc=$?"
cd $R
S=$(find $O/trace -name '*kernel_stats.csv' | head -1); cp "$S" $O/kernel_stats.csv
python - <<PY
import csv
rows=list(csv.DictReader(open("$O/kernel_stats.csv")))
out=sorted(((float(r["TotalDurationNs"])/1e3, r["Name"][:60], int(r["Calls"]), float(r["AverageNs"])/1e3) for r in rows), reverse=True)
for t,n,c,a in out[:28]: print(f"{t/25:8.1f} us/step calls/step {c/25:4.1f} avg {a:7.1f} {n}")
PY
rm -rf $O/trace
