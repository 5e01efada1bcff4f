#!/bin/bash
# Evidence set for profiles/<tag>: GPU tests, the bench line, per-kernel trace stats of the same
# command, PMC traffic passes (separate runs, kernel-trace only).
set -u
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
TAG=${1:-r01e}
R=$PWD; O=$R/gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q > $O/gpu_tests.log 2>&1; grep -E "passed|failed" $O/gpu_tests.log | tail -1
timeout 500 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; cut -c1-200 $O/bench.json
if [ -z "${SKIP_EXTRA:-}" ]; then  # SKIP_EXTRA=1: only the default line (when GPU minutes are short)
timeout 300 python bench.py --global-batch 8192 --no-cpu-baseline > $O/bench_b8192.json 2>> $O/bench.err; echo "bench b8192 rc=$?"
timeout 300 python bench.py --force-sharded --replicate-small --no-cpu-baseline 2>> $O/bench.err | tail -1 > $O/sharded_w1_proxy_bench.json; echo "sharded proxy rc=$?"
# delta-embedding tracker: the default line with every lookup recorded, and the tracker on its own
timeout 300 python bench.py --delta-tracker --no-cpu-baseline > $O/bench_delta_tracker.json 2>> $O/bench.err; echo "bench delta-tracker rc=$?"
timeout 300 python scripts/bench_delta.py > $O/bench_delta.txt 2>> $O/bench.err; echo "bench_delta rc=$?"; cat $O/bench_delta.txt
fi
cd /tmp
# tuning stays on: the shipped table covers every shape of this run, so no candidate kernels appear
timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace -o t --output-format csv -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-graph > $O/trace.log 2>&1; echo "trace rc=$?"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $c -d $O/pmc_$c -o p --output-format csv -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-graph > /dev/null 2>&1; echo "pmc $c rc=$?"
done
cd $R
python scripts/pmc_summary.py $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/pmc_traffic.json
S=$(find $O/trace -name '*kernel_stats.csv' | head -1); cp "$S" $O/kernel_stats.csv
grep tzr_ $O/kernel_stats.csv | cut -c1-60,200-400 | head -20
rm -rf $O/trace $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE
