#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
TAG=${1:-plan}; R=$PWD; O=$R/gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
cd /tmp
timeout 300 rocprofv3 --kernel-trace -d $O/trace -o t -- python $R/scripts/plan_trace.py ${2:-uniform} ${3:-65536} ${4:-adagrad} ${5:-0} > $O/trace.log 2>&1; echo "trace rc=$?"
cd $R
DB=$(find $O/trace -name '*.db' | head -1)
python scripts/rocpd_timeline.py "$DB" 16 $O/timeline.txt
python scripts/rocpd_stats.py "$DB" $O/kernel_stats.csv
rm -rf $O/trace
cat $O/timeline.txt | cut -c1-100
cut -d, -f1-7 $O/kernel_stats.csv | cut -c1-140
