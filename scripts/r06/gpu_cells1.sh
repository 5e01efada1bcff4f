#!/bin/bash
# cells plan, first hardware run: parity tests of the three backward forms, then same-box timing exact vs cells (and the build
# without the in-kernel slow path, to see what its registers cost the tile loop)
set -u
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
TAG=${1:-r06b}
O=$PWD/gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_pooled_parity.py tests/test_fullsize_properties.py -m gpu -q -x > $O/gpu_tests_pooled.log 2>&1; tail -3 $O/gpu_tests_pooled.log
timeout 600 python scripts/emb_ab.py --plan exact,cells --iters 40 > $O/emb_ab.txt 2>&1; cat $O/emb_ab.txt | grep "^B"
timeout 600 python scripts/emb_ab.py --plan cells --iters 40 --lib torcheasyrec_amd/libtzrec_hip_noslow.so > $O/emb_ab_noslow.txt 2>&1; grep "^B" $O/emb_ab_noslow.txt
timeout 600 python scripts/emb_ab.py --plan exact,cells --iters 40 --opt rowwise_adagrad >> $O/emb_ab.txt 2>&1; grep "^B" $O/emb_ab.txt | tail -2
timeout 600 python scripts/emb_ab.py --plan exact,cells,auto --iters 40 --dist zipf > $O/emb_ab_zipf.txt 2>&1; grep "^B" $O/emb_ab_zipf.txt
