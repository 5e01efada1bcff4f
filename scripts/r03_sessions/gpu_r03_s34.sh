#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
TAG=${1:-r03bn}; R=$PWD; O=$R/gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
for v in "" "--async-plan" "" "--async-plan"; do
  timeout 400 python bench.py --no-cpu-baseline --no-e2e --no-secondary $v > $O/bench$v.json 2>> $O/bench.err; python -c "
import json; d=json.load(open('$O/bench$v.json')); print('[$v]', round(d['value']/1e6,2),'M/s', round(d['ms_per_step'],4),'ms loss', d['final_loss'])"
done
for v in "" "--async-plan"; do
  timeout 400 python bench.py --no-cpu-baseline --no-e2e --no-secondary --global-batch 8192 $v > $O/bench8k$v.json 2>> $O/bench.err; python -c "
import json; d=json.load(open('$O/bench8k$v.json')); print('[8192 $v]', round(d['value']/1e6,2),'M/s', round(d['ms_per_step'],4),'ms loss', d['final_loss'])"
done
