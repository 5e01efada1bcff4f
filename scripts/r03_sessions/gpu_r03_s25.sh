#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
TAG=${1:-r03ay}; R=$PWD; O=$R/gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_interaction_top.py -m gpu -q -x > $O/tests.log 2>&1; echo "tests rc=$?"; grep -E "passed|failed" $O/tests.log | tail -1
IT_WGS=0 IT_STAGGER=0,1,2,0,1,2 timeout 300 python scripts/bench_interaction_top.py 65536 > $O/bench_interaction_top.txt 2>&1; grep -v amdgpu.ids $O/bench_interaction_top.txt
