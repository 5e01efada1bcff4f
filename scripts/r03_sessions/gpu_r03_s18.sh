#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
TAG=${1:-r03ad}; R=$PWD; O=$R/gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
TZR_TUNABLE_TUNING=1 IT_WGS=${IT_WGS:-0,128,256} timeout 300 python scripts/bench_interaction_top.py > $O/bench_interaction_top.txt 2>&1; cat $O/bench_interaction_top.txt | grep -v amdgpu.ids
cd /tmp
TZR_TUNABLE_TUNING=0 timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof -o t --output-format csv -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-graph --no-e2e --no-secondary > $O/trace.log 2>&1
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/kernel_stats.csv && head -22 $O/kernel_stats.csv | cut -c1-140
rm -rf $O/prof
