#!/bin/bash
# gpurun -- bash scripts/r06/gpu_din_trace.sh <tag>: kernel table of the multi_tower_din step with the DIN towers on the library's own products
set -u
TAG=${1:-r06s}; R=$PWD; O=$R/gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
cd /tmp
TZR_TUNABLE_TUNING=0 timeout 400 rocprofv3 --kernel-trace -d $O/trace -o t -- python $R/scripts/r05/din_step.py 20 jagged > $O/trace.log 2>&1; echo "trace rc=$?"; grep din_towers $O/trace.log
cd $R
DB=$(find $O/trace -name '*.db' | head -1)
python scripts/rocpd_stats.py "$DB" $O/kernel_stats_din_own.csv
rm -rf $O/trace
head -45 $O/kernel_stats_din_own.csv | cut -d, -f1-7 | cut -c1-150
