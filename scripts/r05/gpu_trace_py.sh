#!/bin/bash
# kernel stats of one python command with the shipped TunableOp selections (no tuning): gpu_trace_py.sh <tag> <name> <python args...>
set -u
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
TAG=$1; NAME=$2; shift 2
R=$PWD; O=$R/gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
cd /tmp
TZR_TUNABLE_TUNING=0 timeout 600 rocprofv3 --kernel-trace -d $O/trace_$NAME -o t -- python "$@" > $O/trace_$NAME.log 2>&1; echo "trace rc=$?"; tail -2 $O/trace_$NAME.log | cut -c1-400
cd $R
DB=$(find $O/trace_$NAME -name '*.db' | head -1)
python scripts/rocpd_stats.py "$DB" $O/kernel_stats_$NAME.csv
rm -rf $O/trace_$NAME
