#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
TAG=${1:-r03bi}; R=$PWD; O=$R/gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_pooled_parity.py tests/test_fullsize_properties.py tests/test_index_parity.py tests/test_sharded_gpu.py -m gpu -q -x > $O/tests.log 2>&1; echo "tests rc=$?"; grep -E "passed|failed" $O/tests.log | tail -1
timeout 200 python scripts/zipf_debug.py 60 > $O/zipf_debug.txt 2>&1; tail -3 $O/zipf_debug.txt
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof -o t --output-format csv -- python $R/scripts/emb_ab.py --dist zipf --B 65536 --iters 20 "" > $O/emb_ab_zipf.txt 2>&1
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/kernel_stats_zipf.csv && grep tzr_bwd $O/kernel_stats_zipf.csv | cut -d, -f1-4 | cut -c1-110
rm -rf $O/prof
