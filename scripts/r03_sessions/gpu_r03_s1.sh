#!/bin/bash
# Round 3, GPU session 1: root-cause experiments for the side-stream plan bug, sharded GPU tests x3 with
# tzr_exchange_pad back in the selection, embedding-path A/B of the new forward / pipelined apply.
set -u
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
TAG=${1:-r03a}; R=$PWD; O=$R/gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
( timeout 120 scripts/probe_xcd_lines.bin 100 > $O/probe_xcd_lines.txt 2>&1; echo "probe rc=$?" )
tail -30 $O/probe_xcd_lines.txt
timeout 900 python scripts/plan_stress.py 150 main main_noise side_idle side_noise side_fwd side_apply main_apply \
  side_apply:bwd_debug=1 side_noise:bwd_debug=1 side_apply:bwd_debug=2 side_noise:bwd_debug=2 side_apply:bwd_debug=4 side_noise:bwd_debug=4 \
  side_apply:bwd_debug=7 side_noise:bwd_debug=7 side_apply:bwd_one_wg_heavy=1 side_noise:bwd_one_wg_heavy=1 > $O/plan_stress.txt 2>&1; echo "stress rc=$?"
grep -E "plan_stress|iter" $O/plan_stress.txt | cut -c1-400 | head -80
timeout 900 python scripts/emb_ab.py --B 65536,8192 --dist uniform,zipf "" "fwd_variant=1" "fwd_tile_b=16" "fwd_tile_b=64" "bwd_apply_pipe=1" "fwd_variant=1,bwd_apply_pipe=1" > $O/emb_ab.txt 2>&1; echo "emb_ab rc=$?"
timeout 300 python scripts/emb_ab.py --opt rowwise_adagrad "" "bwd_apply_pipe=1" >> $O/emb_ab.txt 2>&1
cat $O/emb_ab.txt | cut -c1-330
for i in 1 2 3; do
  timeout 600 python -m pytest tests/test_index_parity.py tests/test_sharded_gpu.py -m gpu -q -x > $O/sharded_tests_$i.log 2>&1; echo "sharded tests $i rc=$?"; tail -2 $O/sharded_tests_$i.log
done
timeout 600 python -m pytest tests/test_pooled_parity.py tests/test_dlrm_parity.py -m gpu -q -x > $O/pooled_tests.log 2>&1; echo "pooled tests rc=$?"; tail -2 $O/pooled_tests.log
timeout 400 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; cut -c1-400 $O/bench.json
