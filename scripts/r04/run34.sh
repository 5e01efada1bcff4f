#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r04at}; mkdir -p $O; export TMPDIR=/tmp
cd /tmp
timeout 200 rocprofv3 --kernel-trace -d $O/trace -o t -- python $R/bench.py --steps 6 --warmup 4 --no-cpu-baseline --no-graph --no-secondary --no-e2e > $O/trace.log 2>&1; echo "trace rc=$?"
cd $R
DB=$(find $O/trace -name '*.db' | head -1)
python scripts/rocpd_timeline.py "$DB" 160 $O/timeline.txt
rm -rf $O/trace
tail -75 $O/timeline.txt | cut -c1-130
