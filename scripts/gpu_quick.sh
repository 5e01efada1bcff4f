#!/bin/bash
# quick GPU iteration: parity tests, bench (B=65536 and 8192), kernel trace summary
set -u
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out; export TMPDIR=/tmp
TAG=${1:-q}
timeout 600 python -m pytest tests -m gpu -q 2>&1 | tail -6
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline ${BENCH_ARGS:-} > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err; echo "bench rc=$?"; python - <<PY
import json
d=json.load(open("gpurun_out/bench_$TAG.json"))
print({k:d[k] for k in ("value","ms_per_step")}, d.get("roofline",{}).get("frac"), d.get("embedding"))
PY
cd /tmp && timeout 300 rocprofv3 --kernel-trace -d "$OLDPWD/gpurun_out/prof_$TAG" -o t -- python "$OLDPWD/bench.py" --steps 10 --warmup 3 --no-cpu-baseline ${BENCH_ARGS:-} > "$OLDPWD/gpurun_out/rocprof_$TAG.log" 2>&1; echo "rocprof rc=$?"; cd "$OLDPWD"
python scripts/rocpd_stats.py gpurun_out/prof_$TAG/t_results.db gpurun_out/kstats_$TAG.csv && head -28 gpurun_out/kstats_$TAG.csv | cut -c1-60,100-400 | awk -F, '{printf "%-62s %6s %9s %10s %6s\n", substr($1,1,62), $2, $3, $6, $7}'
