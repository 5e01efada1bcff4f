#!/bin/bash
# Zipf ids: the sort launch split into its unit and heavy-worker halves (bwd_debug=2), per-kernel trace (raw, per dispatch)
set -u
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
TAG=${1:-r03bg}; R=$PWD; O=$R/gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
cd /tmp
timeout 300 rocprofv3 --kernel-trace -d $O/prof -o t --output-format csv -- python $R/scripts/emb_ab.py --dist zipf --B 65536 --iters 6 "bwd_debug=2" > $O/emb_ab_zipf_split.txt 2>&1
f=$(find $O/prof -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && python - "$f" <<'PY'
import csv, sys
rows=[r for r in csv.DictReader(open(sys.argv[1])) if 'tzr_bwd_sort' in r['Kernel_Name']]
for r in rows[-12:]:
    print(r['Kernel_Name'][:24], 'grid', r.get('Grid_Size_X', r.get('Grid_Size')), 'us', (int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
PY
rm -rf $O/prof; grep "^B " $O/emb_ab_zipf_split.txt
