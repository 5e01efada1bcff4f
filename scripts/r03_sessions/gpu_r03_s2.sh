#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
TAG=${1:-r03b}; R=$PWD; O=$R/gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
timeout 600 python scripts/emb_ab.py --B 65536 "bwd_ch=1024" "bwd_ch=768" "bwd_ch=512" "bwd_ch=256" > $O/emb_ab_ch.txt 2>&1; echo "emb_ab rc=$?"
cat $O/emb_ab_ch.txt | cut -c1-330
timeout 900 python scripts/zipf_debug.py 60 0:0 > $O/zipf_debug_v0.txt 2>&1; echo "zipf_debug rc=$?"
grep -v "^debug 0 variant 0 iter" $O/zipf_debug_v0.txt | cut -c1-300 | tail -20
