#!/bin/bash
# one GPU call: the embedding parity tests on hardware, then the stage A/B of scripts/emb_ab.py under knob sets
set -u
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
TAG=${1:-r05a}; O=gpurun_out/$TAG; mkdir -p $O
timeout 900 python -m pytest tests/test_pooled_parity.py tests/test_plan_invariants.py tests/test_fullsize_properties.py -m gpu -x -q > $O/gpu_tests_embedding.log 2>&1; echo "tests rc=$?"; tail -3 $O/gpu_tests_embedding.log
shift
timeout 600 python scripts/emb_ab.py --B 65536 --dist uniform "$@" > $O/emb_ab_uniform.txt 2>&1; echo "ab rc=$?"; cat $O/emb_ab_uniform.txt | cut -c1-260
timeout 300 python scripts/emb_ab.py --B 65536 --dist zipf "" "bwd_solo=0" > $O/emb_ab_zipf.txt 2>&1; cat $O/emb_ab_zipf.txt | cut -c1-260
timeout 300 python scripts/emb_ab.py --opt rowwise_adagrad --B 65536 "" "bwd_solo=0" > $O/emb_ab_rowwise.txt 2>&1; cat $O/emb_ab_rowwise.txt | cut -c1-260
