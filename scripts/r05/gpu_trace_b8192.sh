#!/bin/bash
# per-kernel trace stats of BASELINE config 2 (global batch 8 192, eager launches, 20 steps) into gpurun_out/<tag>/kernel_stats_b8192.csv
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
TAG=${1:-r05x}; R=$PWD; O=$R/gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace -o t --output-format csv -- python $R/bench.py --global-batch 8192 --steps 20 --warmup 5 --no-cpu-baseline --no-e2e --no-graph --no-secondary > $O/trace.log 2>&1; echo "trace rc=$?"
cd $R
S=$(find $O/trace -name '*kernel_stats.csv' | head -1); cp "$S" $O/kernel_stats_b8192.csv; rm -rf $O/trace
python - <<PY
import csv
rows=[r for r in csv.DictReader(open("$O/kernel_stats_b8192.csv")) if int(r["Calls"])>=20]
tot=0
for r in sorted(rows,key=lambda r:-float(r["TotalDurationNs"]))[:24]:
    c=int(r["Calls"]); per=float(r["TotalDurationNs"])/25 if c%25==0 else float(r["AverageNs"])
    tot+=per
    print(f"{r['Name'][:56]:56s} {r['Calls']:>5s} {float(r['AverageNs'])/1e3:8.1f} {per/1e3:8.1f}")
print("sum per step", tot/1e3)
PY
