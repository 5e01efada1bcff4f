#!/bin/bash
# per-collective launches of a 1-rank RCCL group
mkdir -p gpurun_out/r03bv; cd /tmp; export TMPDIR=/tmp
for w in none a2a a2a_even allreduce; do
  timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/r03bv/$w -- python /root/repo/scripts/probe_rccl_fills.py $w > /root/repo/gpurun_out/r03bv/$w.log 2>&1
  f=$(find /root/repo/gpurun_out/r03bv/$w -name "*kernel_stats.csv" | head -1); echo "== $w"; cut -d, -f1-4 "$f" | cut -c1-110 | head -8
  find /root/repo/gpurun_out/r03bv/$w -name "*.csv" ! -name "*kernel_stats*" -delete
done
