#!/bin/bash
# gpurun -- bash scripts/r06/gpu_mmoe_trace.sh <tag>: kernel table of the MMoE + ZCH step (BASELINE configs[4]) and of DeepFM
set -u
TAG=${1:-r06al}; R=$PWD; O=$R/gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
cd /tmp
for m in mmoe_zch_b8192 deepfm_criteo_b8192; do
TZR_TUNABLE_TUNING=0 timeout 400 rocprofv3 --kernel-trace -d $O/trace_$m -o t -- python $R/scripts/r05/models_step.py 20 $m > $O/trace_$m.log 2>&1; echo "trace $m rc=$?"; grep '"model"' $O/trace_$m.log | cut -c1-200
DB=$(find $O/trace_$m -name '*.db' | head -1)
python $R/scripts/rocpd_stats.py "$DB" $O/kernel_stats_$m.csv
rm -rf $O/trace_$m
done
