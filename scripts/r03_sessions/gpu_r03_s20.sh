#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
TAG=${1:-r03af}; R=$PWD; O=$R/gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_interaction_top.py -m gpu -q -x > $O/tests.log 2>&1; echo "tests rc=$?"; grep -E "passed|failed" $O/tests.log | tail -1
IT_WGS=0 IT_DEBUG=${IT_DEBUG:-} timeout 300 python scripts/bench_interaction_top.py ${BS:-65536,8192} > $O/bench_interaction_top.txt 2>&1; cat $O/bench_interaction_top.txt | grep -v amdgpu.ids
