#!/bin/bash
# One unsharded DLRM step, kernel by kernel (eager launch order = captured graph order).
set -u
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
TAG=${1:-step}; R=$PWD; O=$R/gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
cd /tmp
timeout 300 rocprofv3 --kernel-trace -d $O/trace -o t -- python $R/bench.py --steps 6 --warmup 4 --no-cpu-baseline --no-graph > $O/trace.log 2>&1; echo "trace rc=$?"
cd $R
DB=$(find $O/trace -name '*.db' | head -1)
python scripts/rocpd_timeline.py "$DB" 700 $O/timeline.txt
rm -rf $O/trace
