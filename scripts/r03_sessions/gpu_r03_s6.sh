#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
TAG=${1:-r03h}; R=$PWD; O=$R/gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_pooled_parity.py tests/test_plan_invariants.py tests/test_dlrm_parity.py tests/test_fullsize_properties.py -m gpu -q -x > $O/plan_tests.log 2>&1; echo "plan tests rc=$?"; tail -3 $O/plan_tests.log
timeout 600 python scripts/plan_stress.py 500 main side_apply main_apply > $O/plan_stress.txt 2>&1; echo "stress rc=$?"
grep -E "plan_stress|iter" $O/plan_stress.txt | cut -c1-300 | head -30
timeout 900 python scripts/emb_ab.py --B 65536,8192 --dist uniform,zipf "" "fwd_plan_fuse=0" > $O/emb_ab.txt 2>&1; echo "emb_ab rc=$?"
cat $O/emb_ab.txt | cut -c1-330
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace -o t --output-format csv -- python $R/scripts/plan_trace.py uniform 65536 adagrad > $O/trace.log 2>&1; echo "trace rc=$?"
cd $R
S=$(find $O/trace -name '*kernel_stats.csv' | head -1); cp "$S" $O/kernel_stats.csv; grep tzr_ $O/kernel_stats.csv | cut -d, -f1-4 | cut -c1-60,100-400 | head -12
rm -rf $O/trace
timeout 400 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; cut -c1-300 $O/bench.json; python -c "
import json; d=json.load(open('$O/bench.json')); print(d.get('embedding')); print(d['roofline']['frac'])"
