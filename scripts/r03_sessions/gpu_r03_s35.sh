#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
TAG=${1:-r03bo}; R=$PWD; O=$R/gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
timeout 300 python scripts/bench_sequence.py > $O/bench_sequence.txt 2>&1; grep -v amdgpu.ids $O/bench_sequence.txt | tail -12
timeout 300 python scripts/bench_interaction.py > $O/bench_interaction.txt 2>&1; grep -v amdgpu.ids $O/bench_interaction.txt | tail -14
