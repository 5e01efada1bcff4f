#!/bin/bash
# kernel table of the DeepFM-Criteo step at batch 8192 (52 tables wide + deep, FM, deep MLP)
set -u
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
TAG=${1:-r05ax}; R=$PWD; O=$R/gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
cd /tmp
TZR_TUNABLE_TUNING=0 timeout 400 rocprofv3 --kernel-trace -d $O/trace -o t -- python $R/scripts/r05/models_step.py 20 deepfm_criteo_b8192 > $O/trace.log 2>&1; echo "trace rc=$?"; grep '"model"' $O/trace.log | cut -c1-300
cd $R
DB=$(find $O/trace -name '*.db' | head -1)
python scripts/rocpd_stats.py "$DB" $O/kernel_stats_deepfm.csv
rm -rf $O/trace
python - $O/kernel_stats_deepfm.csv <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:32]:
    print(r['kernel'][:70].ljust(70), r['calls'].rjust(6), r['avg_us'].rjust(9), r['total_us'].rjust(10), r['pct'].rjust(6))
PY
