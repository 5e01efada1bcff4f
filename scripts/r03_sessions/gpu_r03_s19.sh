#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
TAG=${1:-r03ae}; R=$PWD; O=$R/gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
IT_WGS=0 IT_DEBUG=${IT_DEBUG:-1,2,4,3,5,6,7} timeout 300 python scripts/bench_interaction_top.py ${BS:-65536} > $O/bench_interaction_top.txt 2>&1; cat $O/bench_interaction_top.txt | grep -v amdgpu.ids
