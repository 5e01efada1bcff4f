#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
TAG=${1:-r03be}; R=$PWD; O=$R/gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
for i in 1 2 3; do
timeout 1500 python -m pytest tests -m gpu -q -x > $O/gpu_tests_$i.log 2>&1; echo "run $i rc=$?"; grep -E "passed|failed" $O/gpu_tests_$i.log | tail -1
done
timeout 400 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err; python -c "
import json; d=json.load(open('$O/bench.json')); print(round(d['value']/1e6,2), d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic'])"
