#!/bin/bash
# cells plan: parity of the three backward forms on hardware, same-box timing exact vs cells, and the build without the worker role
set -u
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
TAG=${1:-r06c}
O=$PWD/gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_pooled_parity.py tests/test_fullsize_properties.py -m gpu -q > $O/gpu_tests_pooled.log 2>&1; tail -3 $O/gpu_tests_pooled.log
timeout 600 python scripts/emb_ab.py --plan exact,cells --iters 40 > $O/emb_ab.txt 2>&1; grep "^B" $O/emb_ab.txt
for v in ${VARIANTS:-noworker}; do
timeout 600 python scripts/emb_ab.py --plan cells --iters 40 --lib torcheasyrec_amd/libtzrec_hip_$v.so > $O/emb_ab_$v.txt 2>&1; echo $v; grep "^B" $O/emb_ab_$v.txt
done
timeout 600 python scripts/emb_ab.py --plan exact,cells --iters 40 --opt rowwise_adagrad >> $O/emb_ab.txt 2>&1; grep "^B" $O/emb_ab.txt | tail -2
timeout 600 python scripts/emb_ab.py --plan exact,cells,auto --iters 40 --dist zipf > $O/emb_ab_zipf.txt 2>&1; grep "^B" $O/emb_ab_zipf.txt
