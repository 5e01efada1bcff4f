#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
TAG=${1:-r04b}
O=gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
timeout 600 python scripts/capture_stress.py 40 > $O/capture_stress.txt 2>&1; echo "stress rc=$?"; grep -v "^frame\|^$" $O/capture_stress.txt | head -40
