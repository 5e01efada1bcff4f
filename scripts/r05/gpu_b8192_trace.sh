#!/bin/bash
# the batch-8192 step (BASELINE configs[1]) kernel by kernel
set -u
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
TAG=${1:-r05af}; R=$PWD; O=$R/gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
cd /tmp
timeout 300 rocprofv3 --kernel-trace -d $O/trace -o t -- python $R/bench.py --global-batch 8192 --steps 6 --warmup 4 --no-cpu-baseline --no-e2e --no-secondary --no-graph > $O/trace.log 2>&1; echo "trace rc=$?"
cd $R
DB=$(find $O/trace -name '*.db' | head -1)
python scripts/rocpd_timeline.py "$DB" 300 $O/timeline_b8192_step.txt
rm -rf $O/trace
python - $O/timeline_b8192_step.txt <<'PY'
import sys
lines=open(sys.argv[1]).read().splitlines()
idx=[i for i,l in enumerate(lines) if 'adam_apply' in l]
i1=idx[-1]; i0=idx[-2]
tot=0
for l in lines[i0+1:i1+1]:
    tot+=float(l.split()[1]); print(l[:120])
print('sum dur',tot)
PY
