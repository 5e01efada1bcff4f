#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
TAG=${1:-r03ak}; R=$PWD; O=$R/gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
IT_WGS=0 timeout 300 python scripts/bench_interaction_top.py 65536 > $O/bench_interaction_top.txt 2>&1; grep -v amdgpu.ids $O/bench_interaction_top.txt
timeout 300 python scripts/bench_interaction_top.py --prof > $O/prof.txt 2>&1; grep -v amdgpu.ids $O/prof.txt
