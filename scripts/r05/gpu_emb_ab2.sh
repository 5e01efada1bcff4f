#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
TAG=${1:-r05c}; O=gpurun_out/$TAG; mkdir -p $O
timeout 300 python scripts/emb_ab.py --lib torcheasyrec_amd/libtzrec_hip_head.so --B 65536 --dist uniform "" "" > $O/emb_ab_head.txt 2>&1; cat $O/emb_ab_head.txt | cut -c1-250
shift
timeout 600 python scripts/emb_ab.py --B 65536 --dist uniform "$@" > $O/emb_ab_new.txt 2>&1; cat $O/emb_ab_new.txt | cut -c1-250
timeout 300 python scripts/emb_ab.py --lib torcheasyrec_amd/libtzrec_hip_head.so --B 65536 --dist uniform "" > $O/emb_ab_head2.txt 2>&1; cat $O/emb_ab_head2.txt | cut -c1-250
