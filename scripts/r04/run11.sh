#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
TAG=${1:-r04t}
O=gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
R=$PWD
cd /tmp
timeout 400 rocprofv3 --kernel-trace -d "$R/$O/trace" -o t -- python "$R/bench.py" --force-sharded --replicate-small --global-batch 8192 --steps 20 --warmup 14 --no-cpu-baseline > "$R/$O/trace.log" 2>&1; echo "trace rc=$?"
cd "$R"
DB=$(find $O/trace -name '*.db' | head -1)
python scripts/rocpd_stats.py "$DB" $O/kernel_stats_proxy8192.csv
python scripts/rocpd_timeline.py "$DB" 200 $O/timeline_proxy8192.txt
rm -rf $O/trace
