#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
TAG=${1:-r03t}; R=$PWD; O=$R/gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_mlp_fused.py tests/test_dlrm_parity.py tests/test_sharded_gpu.py tests/test_graph_pipeline_gpu.py -m gpu -q -x > $O/tests.log 2>&1; echo "tests rc=$?"; grep -E "passed|failed|Error" $O/tests.log | tail -3
for v in "" "--layerwise-loss" ; do
  timeout 400 python bench.py --no-cpu-baseline --no-e2e --no-secondary $v > $O/bench$v.json 2>> $O/bench.err; python -c "
import json; d=json.load(open('$O/bench$v.json')); print('[$v]', round(d['value']/1e6,2),'M/s', round(d['ms_per_step'],4),'ms loss', d['final_loss'])"
done
TZR_FUSED_MLP2=0 timeout 400 python bench.py --no-cpu-baseline --no-e2e --no-secondary > $O/bench_nomlp2.json 2>> $O/bench.err; python -c "
import json; d=json.load(open('$O/bench_nomlp2.json')); print('[no mlp2]', round(d['value']/1e6,2),'M/s', round(d['ms_per_step'],4),'ms loss', d['final_loss'])"
timeout 400 python bench.py --no-cpu-baseline --no-e2e --no-secondary --global-batch 8192 > $O/bench_b8192.json 2>> $O/bench.err; python -c "
import json; d=json.load(open('$O/bench_b8192.json')); print('[8192]', round(d['value']/1e6,2),'M/s', round(d['ms_per_step'],4),'ms loss', d['final_loss'])"
timeout 400 python bench.py --no-cpu-baseline --no-e2e --no-secondary --global-batch 8192 --layerwise-loss > $O/bench_b8192_lw.json 2>> $O/bench.err; python -c "
import json; d=json.load(open('$O/bench_b8192_lw.json')); print('[8192 layerwise]', round(d['value']/1e6,2),'M/s', round(d['ms_per_step'],4),'ms loss', d['final_loss'])"
tail -3 $O/bench.err
