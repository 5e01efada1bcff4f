#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
TAG=${1:-r03j}; R=$PWD; O=$R/gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_pooled_parity.py tests/test_plan_invariants.py -m gpu -q -x > $O/plan_tests.log 2>&1; echo "plan tests rc=$?"; tail -1 $O/plan_tests.log
timeout 900 python scripts/emb_ab.py --B 65536,8192 --dist uniform "" "fwd_plan_mix=2" "fwd_plan_mix=3" "fwd_plan_mix=5" "fwd_plan_mix=8" "fwd_plan_fuse=0" > $O/emb_ab.txt 2>&1; echo "emb_ab rc=$?"
cat $O/emb_ab.txt | cut -c1-330
