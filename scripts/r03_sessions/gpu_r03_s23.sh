#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
TAG=${1:-r03as}; R=$PWD; O=$R/gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
for v in 0 -1; do
TZR_TUNE=mlp_mfma=$v timeout 300 python -m pytest tests/test_sharded_gpu.py -m gpu -q -x -k "whole_step_graph" > $O/tests_mfma$v.log 2>&1; echo "mlp_mfma=$v rc=$?"; grep -E "passed|failed|Error|assert|Fatal" $O/tests_mfma$v.log | head -8
done
TZR_FUSED_IA_TOP=0 timeout 300 python -m pytest tests/test_sharded_gpu.py -m gpu -q -x -k "whole_step_graph" > $O/tests_nofuse.log 2>&1; echo "nofuse rc=$?"; grep -E "passed|failed|Error|assert|Fatal" $O/tests_nofuse.log | head -8
