#!/bin/bash
# heavy-bucket walks by wave-sized segments: plan parity on the GPU (incl. the full-size Zipf checks), Zipf / uniform timing
set -u
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
TAG=${1:-r03bh}; R=$PWD; O=$R/gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_pooled_parity.py tests/test_fullsize_properties.py tests/test_sharded_gpu.py tests/test_index_parity.py -m gpu -q -x > $O/tests.log 2>&1; echo "tests rc=$?"; grep -E "passed|failed" $O/tests.log | tail -1
timeout 300 python scripts/emb_ab.py --dist zipf,uniform --B 65536,8192 --iters 20 "" > $O/emb_ab.txt 2>&1; grep "^B " $O/emb_ab.txt
timeout 200 python scripts/plan_stress.py 300 side_apply main > $O/plan_stress.txt 2>&1; tail -3 $O/plan_stress.txt
