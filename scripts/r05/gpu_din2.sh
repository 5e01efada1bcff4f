#!/bin/bash
# phase 1: the DIN step with TunableOp tuning, table saved; phase 2: kernel trace of the jagged step with the saved selections, no tuning
set -u
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
TAG=${1:-r05f}; R=$PWD; O=$R/gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
TZR_TUNABLE_SAVE=$O/tunableop_tuned.csv timeout 900 python scripts/r05/din_step.py 30 ${2:-both} > $O/din_step.txt 2>&1; grep din_towers $O/din_step.txt || tail -20 $O/din_step.txt
if [ -s $O/tunableop_tuned.csv ]; then cp $O/tunableop_tuned.csv torcheasyrec_amd/tunableop_gfx950.csv; fi
wc -l torcheasyrec_amd/tunableop_gfx950.csv
cd /tmp
TZR_TUNABLE_TUNING=0 timeout 400 rocprofv3 --kernel-trace -d $O/trace -o t -- python $R/scripts/r05/din_step.py 20 jagged > $O/trace.log 2>&1; echo "trace rc=$?"; grep din_towers $O/trace.log
cd $R
DB=$(find $O/trace -name '*.db' | head -1)
python scripts/rocpd_stats.py "$DB" $O/kernel_stats_din_jagged.csv
rm -rf $O/trace
head -42 $O/kernel_stats_din_jagged.csv | cut -d, -f1-7 | cut -c1-200
