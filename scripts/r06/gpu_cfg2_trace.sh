#!/bin/bash
# kernel table of the eager step at global batch $1 (default 8192 = config 2)
R=$PWD; O=$R/gpurun_out/cfg2trace; mkdir -p $O; export TMPDIR=/tmp; cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/t -o t --output-format csv -- python $R/bench.py --global-batch ${1:-8192} --steps 20 --warmup 5 --no-cpu-baseline --no-e2e --no-graph --no-secondary > $O/log 2>&1
S=$(find $O/t -name '*kernel_stats.csv' | head -1)
python - "$S" > $O/kernels.txt <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'tzr_' in r['Name'] and int(r['Calls']) >= 20:
        print(r['Name'].split('(')[0][-44:].ljust(46), r['Calls'].rjust(5), '%9.1f' % (float(r['AverageNs']) / 1e3), 'us  min %8.1f' % (float(r['MinNs']) / 1e3))
PY
rm -rf $O/t; cat $O/kernels.txt
