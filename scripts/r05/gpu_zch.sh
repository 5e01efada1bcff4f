#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
TAG=${1:-r05i}; R=$PWD; O=$R/gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_graph_pipeline_gpu.py tests/test_zch_parity.py -m gpu -x -q > $O/gpu_tests_zch.log 2>&1; echo "tests rc=$?"; tail -5 $O/gpu_tests_zch.log
TZR_TUNABLE_SAVE=$O/tunableop_tuned.csv timeout 900 python scripts/r05/zch_step.py 30 > $O/zch_step.txt 2>&1; tail -3 $O/zch_step.txt | cut -c1-600
