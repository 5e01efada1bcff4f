#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 120 ./scripts/probe_hbm.bin | tee gpurun_out/probe_hbm.txt
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 200 rocprofv3 --kernel-trace --pmc $c -d "$OLDPWD/gpurun_out/probe_pmc_$c" -o p --output-format csv -- "$OLDPWD/scripts/probe_hbm.bin" > /dev/null 2>&1; echo "pmc $c rc=$?"
done
cd "$OLDPWD"
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/bench_tun.json 2> gpurun_out/bench_tun.err; echo "bench rc=$?"; tail -2 gpurun_out/bench_tun.err; python -c "
import json; d=json.load(open('gpurun_out/bench_tun.json')); print(d['value'], d['ms_per_step'], d['final_loss'])"
