#!/bin/bash
# kernel trace of the 8192-per-rank sharded step (1-rank RCCL group, native driver): kernel time per step vs the step's wall time
set -u
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
TAG=${1:-r05q}; R=$PWD; O=$R/gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
cd /tmp
timeout 400 rocprofv3 --kernel-trace -d $O/trace -o t -- python $R/bench.py --gpus 1 --force-sharded --replicate-small --global-batch 8192 --steps 40 --warmup 12 --no-cpu-baseline --no-e2e --projection-world 8 > $O/trace.log 2>&1; echo "trace rc=$?"
cd $R
grep "^{" $O/trace.log | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ms_per_step',d['ms_per_step'],'host_queue',d['host_queue_ms_per_step'])"
DB=$(find $O/trace -name '*.db' | head -1)
python scripts/rocpd_stats.py "$DB" $O/kernel_stats.csv
python scripts/rocpd_timeline.py "$DB" 130 $O/timeline.txt
rm -rf $O/trace
cut -c1-150 $O/timeline.txt | tail -135
