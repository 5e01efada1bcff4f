#!/bin/bash
# per-kernel times of the embedding path on Zipf ids
set -u
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
TAG=${1:-r03bf}; R=$PWD; O=$R/gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof -o t --output-format csv -- python $R/scripts/emb_ab.py --dist zipf --B 65536 --iters 20 "" > $O/emb_ab_zipf.txt 2>&1
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/kernel_stats_zipf.csv && grep tzr_ $O/kernel_stats_zipf.csv | cut -d, -f1-4 | cut -c1-120
rm -rf $O/prof; grep "^B " $O/emb_ab_zipf.txt
