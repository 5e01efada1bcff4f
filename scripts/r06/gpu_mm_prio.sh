#!/bin/bash
# (RECORD ONLY if removed) wave priorities by residency slot in the three small MFMA stacks (mlp_mfma.hip, -DMM_PRIO): kernel table, product vs variant
R=$PWD; O=$R/gpurun_out/mmprio; mkdir -p $O; export TMPDIR=/tmp; cd /tmp
for rep in 1 2; do
for v in product mmprio; do
  F=""; [ $v = mmprio ] && F="--lib libtzrec_hip_mmprio.so"
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/$v -o t --output-format csv -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-e2e --no-graph --no-secondary $F > $O/$v.log 2>&1
  S=$(find $O/$v -name '*kernel_stats.csv' | head -1)
  echo "== $v"; python - "$S" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'tzr_mlp' in r['Name'] and int(r['Calls']) >= 20:
        print(r['Name'].split('(')[0][-30:].ljust(32), '%7.2f' % (float(r['AverageNs']) / 1e3), 'us  min %7.2f' % (float(r['MinNs']) / 1e3))
PY
  rm -rf $O/$v
done
done
