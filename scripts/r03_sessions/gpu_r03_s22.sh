#!/bin/bash
# MFMA kernels for the small MLP stacks: parity on the GPU, bench A/B against the LDS-tiled kernels, kernel stats
set -u
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
TAG=${1:-r03ar}; R=$PWD; O=$R/gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_interaction_top.py tests/test_mlp_fused.py tests/test_sharded_gpu.py tests/test_graph_pipeline_gpu.py -m gpu -q -x > $O/tests.log 2>&1; echo "tests rc=$?"; grep -E "passed|failed" $O/tests.log | tail -1
for v in 0 -1 ; do
  TZR_TUNE=mlp_mfma=$v timeout 400 python bench.py --no-cpu-baseline --no-e2e --no-secondary > $O/bench_mfma$v.json 2>> $O/bench.err; python -c "
import json; d=json.load(open('$O/bench_mfma$v.json')); print('[mlp_mfma=$v]', round(d['value']/1e6,2),'M/s', round(d['ms_per_step'],4),'ms loss', d['final_loss'])"
  TZR_TUNE=mlp_mfma=$v timeout 400 python bench.py --no-cpu-baseline --no-e2e --no-secondary --global-batch 8192 > $O/bench8k_mfma$v.json 2>> $O/bench.err; python -c "
import json; d=json.load(open('$O/bench8k_mfma$v.json')); print('[8192 mlp_mfma=$v]', round(d['value']/1e6,2),'M/s', round(d['ms_per_step'],4),'ms loss', d['final_loss'])"
done
cd /tmp
TZR_TUNABLE_TUNING=0 timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof -o trace --output-format csv -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-graph --no-e2e --no-secondary > $O/trace.log 2>&1
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/kernel_stats.csv && head -24 $O/kernel_stats.csv | cut -c1-110
rm -rf $O/prof
