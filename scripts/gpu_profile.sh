#!/bin/bash
# Record set for profiles/: GPU tests, bench JSON, kernel trace, PMC passes (separate runs).
set -u
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
TAG=${1:-r01}
mkdir -p gpurun_out/$TAG; export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed" | tail -2
timeout 400 python bench.py --steps 30 --warmup 5 > gpurun_out/$TAG/bench.json 2> gpurun_out/$TAG/bench.err; echo "bench rc=$?"
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d "$OLDPWD/gpurun_out/$TAG/trace" -o t --output-format csv -- python "$OLDPWD/bench.py" --steps 10 --warmup 3 --no-cpu-baseline --no-graph > "$OLDPWD/gpurun_out/$TAG/trace.log" 2>&1; echo "trace rc=$?"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $c -d "$OLDPWD/gpurun_out/$TAG/pmc_$c" -o p --output-format csv -- python "$OLDPWD/bench.py" --steps 4 --warmup 2 --no-cpu-baseline --no-graph > /dev/null 2>&1; echo "pmc $c rc=$?"
done
cd "$OLDPWD"; ls gpurun_out/$TAG/trace | head
