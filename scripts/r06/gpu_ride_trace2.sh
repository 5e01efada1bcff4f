#!/bin/bash
# (RECORD ONLY: tzr_pooled_bwd_cells_apply_adam / bench.py --no-ride were built, measured -- profiles/r06ax -- and removed in round 6)
# diagnosis of the ride: rider workgroups that return at once / that run at priority 3
R=$PWD; O=$R/gpurun_out/ridetrace2; mkdir -p $O; export TMPDIR=/tmp; cd /tmp
for v in ridernop riderprio; do
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/$v -o t --output-format csv -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-e2e --no-graph --no-secondary --lib libtzrec_hip_$v.so > $O/$v.log 2>&1
  S=$(find $O/$v -name '*kernel_stats.csv' | head -1)
  python - "$S" > $O/$v.txt <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'tzr_bwd_cells' in r['Name'] or 'adam' in r['Name']:
        print(r['Name'].split('(')[0][-44:].ljust(46), r['Calls'].rjust(5), '%9.1f' % (float(r['AverageNs']) / 1e3), 'us  min %8.1f' % (float(r['MinNs']) / 1e3))
PY
  rm -rf $O/$v
  echo "== $v"; cat $O/$v.txt
done
