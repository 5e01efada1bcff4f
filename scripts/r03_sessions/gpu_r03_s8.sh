#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
TAG=${1:-r03l}; R=$PWD; O=$R/gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_pooled_parity.py tests/test_plan_invariants.py -m gpu -q -x > $O/plan_tests.log 2>&1; echo "plan tests rc=$?"; tail -1 $O/plan_tests.log
timeout 900 python scripts/emb_ab.py --B 65536,8192 --dist uniform,zipf "" "bwd_pk=1" "bwd_pk=2" "fwd_plan_fuse=0" "fwd_plan_fuse=0,bwd_pk=1" > $O/emb_ab.txt 2>&1; echo "emb_ab rc=$?"
cat $O/emb_ab.txt | cut -c1-330
python scripts/part_prof.py 65536 1 2>&1 | tail -14
python scripts/part_prof.py 65536 0 2>&1 | tail -14
