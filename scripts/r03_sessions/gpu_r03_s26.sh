#!/bin/bash
# sharded step on the 1-rank RCCL proxy at batch 8192: bench line + per-kernel trace
set -u
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
TAG=${1:-r03az}; R=$PWD; O=$R/gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
for gb in 8192 65536; do
timeout 300 python bench.py --force-sharded --replicate-small --no-cpu-baseline --global-batch $gb > $O/bench_$gb.out 2> $O/bench_$gb.err; echo "bench $gb rc=$?"; tail -1 $O/bench_$gb.out | cut -c1-250
cd /tmp
TZR_TUNABLE_TUNING=0 timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_$gb -o t --output-format csv -- python $R/bench.py --force-sharded --replicate-small --steps 20 --warmup 5 --no-cpu-baseline --global-batch $gb > $O/trace_$gb.log 2>&1; echo "trace rc=$?"
cd $R
f=$(find $O/prof_$gb -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/kernel_stats_$gb.csv
rm -rf $O/prof_$gb
done
