#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
TAG=${1:-r03n}; R=$PWD; O=$R/gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
timeout 600 python scripts/emb_ab.py --B 65536,8192 "" "bwd_apply_waves=7" "bwd_apply_waves=8" > $O/emb_ab.txt 2>&1
timeout 600 python scripts/emb_ab.py --opt rowwise_adagrad --layout interleaved "" "bwd_apply_waves=7" >> $O/emb_ab.txt 2>&1
timeout 600 python scripts/emb_ab.py --opt rowwise_adagrad --layout split "" >> $O/emb_ab.txt 2>&1
cat $O/emb_ab.txt | cut -c1-330
timeout 600 python -m pytest tests/test_pooled_parity.py tests/test_fullsize_properties.py -m gpu -q -x > $O/tests.log 2>&1; echo "tests rc=$?"; tail -1 $O/tests.log
