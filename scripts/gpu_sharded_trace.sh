#!/bin/bash
# Kernel trace of the sharded step on a 1-rank RCCL group (all exchange kernels run; a2a is a self copy).
set -u
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
TAG=${1:-shard}
mkdir -p gpurun_out/$TAG; export TMPDIR=/tmp
R=$PWD
timeout 300 python bench.py --force-sharded --replicate-small --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/$TAG/bench.json 2> gpurun_out/$TAG/bench.err; echo "bench rc=$?"; cat gpurun_out/$TAG/bench.json | cut -c1-300
cd /tmp
timeout 300 rocprofv3 --kernel-trace -d "$R/gpurun_out/$TAG/trace" -o t -- python "$R/bench.py" --force-sharded --replicate-small --steps 10 --warmup 3 --no-cpu-baseline > "$R/gpurun_out/$TAG/trace.log" 2>&1; echo "trace rc=$?"
cd "$R"
DB=$(find gpurun_out/$TAG/trace -name '*.db' | head -1)
python scripts/rocpd_stats.py "$DB" gpurun_out/$TAG/kernel_stats.csv
python scripts/rocpd_timeline.py "$DB" 150 gpurun_out/$TAG/timeline.txt
rm -rf gpurun_out/$TAG/trace
