#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
TAG=${1:-r03o}; R=$PWD; O=$R/gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
timeout 600 python scripts/emb_ab.py --B 65536 --dist uniform,zipf "" "fwd_tile_b=40" "fwd_tile_b=44" "fwd_tile_b=48" "fwd_tile_b=64" "fwd_tile_b=24" > $O/emb_ab.txt 2>&1
cat $O/emb_ab.txt | cut -c1-330
