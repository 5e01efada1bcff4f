#!/bin/bash
# rows gather with batched loads / small workgroups: sharded GPU tests, per-kernel trace of the 8192 proxy
mkdir -p gpurun_out/r03bv
timeout 600 python -m pytest tests/test_sharded_gpu.py tests/test_zz_mixed_sharded_world1.py -q -m gpu -x > gpurun_out/r03bv/pytest_sharded.txt 2>&1; tail -2 gpurun_out/r03bv/pytest_sharded.txt
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/r03bv/trace8192 -- python /root/repo/bench.py --force-sharded --replicate-small --no-cpu-baseline --global-batch 8192 > /root/repo/gpurun_out/r03bv/sharded_8192_traced.json 2>/dev/null
cd /root/repo
f=$(find gpurun_out/r03bv/trace8192 -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/r03bv/kernel_stats_8192.csv; grep -E "rows_gather|lookup_grads|xb_" gpurun_out/r03bv/kernel_stats_8192.csv | cut -d, -f1-4 | cut -c1-60,200-
find gpurun_out/r03bv/trace8192 -name "*.csv" -delete
tail -1 gpurun_out/r03bv/sharded_8192_traced.json | cut -c1-260
