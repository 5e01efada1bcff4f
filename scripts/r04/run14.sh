#!/bin/bash
mkdir -p gpurun_out/r04x
WG_DEBUG=1,2,4,6,7,3 timeout 300 python scripts/bench_interaction_top.py 65536 > gpurun_out/r04x/bench_wgrad_phases.txt 2>&1
grep "wgrad" gpurun_out/r04x/bench_wgrad_phases.txt
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof -o wg -- python $GRAFT_REPO_ROOT/scripts/bench_interaction_top.py 65536 > /dev/null 2>&1
f=$(find /tmp/prof -name "*kernel_stats.csv" | head -1); grep "wgrad" $f | cut -c1-160 | tee $GRAFT_REPO_ROOT/gpurun_out/r04x/kernel_stats_wgrad.txt
