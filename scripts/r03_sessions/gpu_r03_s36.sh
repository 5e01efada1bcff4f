#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
TAG=${1:-r03bp}; R=$PWD; O=$R/gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof -o t --output-format csv -- python $R/scripts/bench_sequence.py > $O/bench_sequence.txt 2>&1
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/kernel_stats_sequence.csv && grep tzr_ $O/kernel_stats_sequence.csv | cut -d, -f1-4 | cut -c1-120
rm -rf $O/prof
