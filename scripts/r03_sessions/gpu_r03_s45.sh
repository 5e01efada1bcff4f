#!/bin/bash
# default bench line with the digest-matched PMC file in place; the 65536 sharded proxy again (host-bound: box variance?)
mkdir -p gpurun_out/r03bu
timeout 500 python bench.py > gpurun_out/r03bu/bench.json 2> gpurun_out/r03bu/bench2.err; cut -c1-400 gpurun_out/r03bu/bench.json
for i in 1 2; do timeout 300 python bench.py --force-sharded --replicate-small --no-cpu-baseline 2>> gpurun_out/r03bu/bench2.err | tail -1 > gpurun_out/r03bu/sharded_w1_proxy_bench_again$i.json; cut -c1-200 gpurun_out/r03bu/sharded_w1_proxy_bench_again$i.json; done
nproc; cat /proc/cpuinfo | grep "model name" | head -1; cat /proc/loadavg
