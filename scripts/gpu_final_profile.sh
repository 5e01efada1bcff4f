#!/bin/bash
# Evidence set for profiles/<tag>: GPU tests, the bench lines (default = global batch 65536 uniform
# Adagrad; batch 8192; row-wise Adagrad; Zipf ids), per-kernel trace stats of the default command,
# PMC passes (separate runs, kernel-trace only): HBM traffic of the six embedding launches and the
# MFMA activity of the interaction kernels.
#   SKIP_TESTS=1  no pytest      SKIP_EXTRA=1  only the default bench line      SKIP_PMC=1  no counter passes
set -u
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
TAG=${1:-r02}
R=$PWD; O=$R/gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
if [ -z "${SKIP_TESTS:-}" ]; then
timeout 1500 python -m pytest tests -m gpu -q --durations=15 > $O/gpu_tests.log 2>&1; grep -E "passed|failed" $O/gpu_tests.log | tail -1
fi
: > $O/bench.err
if [ -z "${SKIP_EXTRA:-}" ]; then
timeout 300 python bench.py --global-batch 8192 --no-cpu-baseline > $O/bench_b8192.json 2>> $O/bench.err; echo "bench b8192 rc=$?"
timeout 300 python bench.py --optimizer rowwise_adagrad --no-cpu-baseline --no-e2e > $O/bench_rowwise_adagrad.json 2>> $O/bench.err; echo "bench rowwise rc=$?"
timeout 300 python bench.py --dist zipf --no-cpu-baseline --no-e2e > $O/bench_zipf.json 2>> $O/bench.err; echo "bench zipf rc=$?"
# 1-rank RCCL proxies of the sharded step: the default (--exchange auto: exact at 65536, capacity + step graphs at 8192) and the exact exchange at 8192
timeout 300 python bench.py --force-sharded --replicate-small --no-cpu-baseline 2>> $O/bench.err | tail -1 > $O/sharded_w1_proxy_bench.json; echo "sharded proxy rc=$?"
timeout 300 python bench.py --force-sharded --replicate-small --no-cpu-baseline --global-batch 8192 2>> $O/bench.err | tail -1 > $O/sharded_w1_proxy_bench_b8192.json; echo "sharded proxy b8192 rc=$?"
timeout 300 python bench.py --force-sharded --replicate-small --no-cpu-baseline --global-batch 8192 --exchange exact 2>> $O/bench.err | tail -1 > $O/sharded_w1_proxy_bench_b8192_exact.json; echo "sharded proxy b8192 exact rc=$?"
fi
cd /tmp
# tuning stays on: the shipped table covers every shape of this run, so no candidate kernels appear
timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace -o t --output-format csv -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-e2e --no-graph --no-secondary > $O/trace.log 2>&1; echo "trace rc=$?"
if [ -z "${SKIP_PMC:-}" ]; then
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $c -d $O/pmc_$c -o p --output-format csv -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-e2e --no-graph --no-secondary > /dev/null 2>&1; echo "pmc $c rc=$?"
done
if [ -z "${SKIP_MFMA:-}" ]; then
# MFMA activity of the dot-interaction kernels (north star: "MFMA utilisation for the interaction against the chip's peak")
rocprofv3 -L 2>/dev/null | grep -i -E "mfma|GRBM_GUI_ACTIVE|SQ_BUSY_CYC" | cut -c1-160 | head -40 > $O/pmc_mfma_counters_available.txt
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $O/pmc_mfma -o p --output-format csv -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-e2e --no-graph --no-secondary > $O/pmc_mfma.log 2>&1; rc=$?; echo "pmc mfma rc=$rc"
if [ $rc -ne 0 ]; then  # a counter name this rocprofv3 does not know: the two that exist everywhere
  rm -rf $O/pmc_mfma
  timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE -d $O/pmc_mfma -o p --output-format csv -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-e2e --no-graph --no-secondary >> $O/pmc_mfma.log 2>&1; echo "pmc mfma (reduced) rc=$?"
fi
fi
fi
cd $R
if [ -z "${SKIP_PMC:-}" ]; then
python scripts/pmc_summary.py $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/pmc_traffic.json
[ -z "${SKIP_MFMA:-}" ] && { python scripts/pmc_mfma_summary.py $O/pmc_mfma $O/pmc_mfma.json || tail -5 $O/pmc_mfma.log; }
fi
# the default line LAST: `roofline.traffic` is read from profiles/r*/pmc_traffic.json of this library digest -- the file just made
if [ -z "${SKIP_PMC:-}" ] && [ -f $O/pmc_traffic.json ]; then mkdir -p profiles/$TAG; cp $O/pmc_traffic.json profiles/$TAG/; fi
timeout 1200 python bench.py > $O/bench.json 2>> $O/bench.err; echo "bench rc=$?"; cut -c1-300 $O/bench.json
S=$(find $O/trace -name '*kernel_stats.csv' | head -1); cp "$S" $O/kernel_stats.csv
grep tzr_ $O/kernel_stats.csv | cut -c1-60,200-400 | head -24
rm -rf $O/trace $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/pmc_mfma
