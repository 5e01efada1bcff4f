#!/bin/bash
# scan launch in slices: parity on the GPU, sequence-path timing, headline embedding path unchanged?
set -u
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
TAG=${1:-r03br}; R=$PWD; O=$R/gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_pooled_parity.py tests/test_sequence_parity.py tests/test_fullsize_properties.py tests/test_index_parity.py -m gpu -q -x > $O/tests.log 2>&1; echo "tests rc=$?"; grep -E "passed|failed" $O/tests.log | tail -1
timeout 300 python scripts/emb_ab.py --dist uniform,zipf --B 65536,8192 --iters 20 "" > $O/emb_ab.txt 2>&1; grep "^B " $O/emb_ab.txt
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof -o t --output-format csv -- python $R/scripts/bench_sequence.py > $O/bench_sequence.txt 2>&1
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/kernel_stats_sequence.csv && grep tzr_bwd $O/kernel_stats_sequence.csv | cut -d, -f1-4 | cut -c1-110
rm -rf $O/prof; grep sparse_backward $O/bench_sequence.txt | cut -c1-400
