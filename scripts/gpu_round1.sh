#!/bin/bash
# First GPU trip: smoke, GPU parity tests, bench (two row layouts), rocprof kernel trace.
set -u
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== smoke"; timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?"; tail -3 gpurun_out/smoke.log
echo "== pytest gpu"; timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -15 gpurun_out/pytest_gpu.log
echo "== bench interleaved"; timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_interleaved.json 2> gpurun_out/bench_interleaved.err; echo "rc=$?"; cat gpurun_out/bench_interleaved.json; tail -5 gpurun_out/bench_interleaved.err
echo "== bench split"; timeout 600 python bench.py --steps 20 --warmup 5 --row-layout split --no-cpu-baseline > gpurun_out/bench_split.json 2> gpurun_out/bench_split.err; echo "rc=$?"; cat gpurun_out/bench_split.json; tail -5 gpurun_out/bench_split.err
echo "== bench B=8192"; timeout 600 python bench.py --steps 20 --warmup 5 --global-batch 8192 --no-cpu-baseline > gpurun_out/bench_b8192.json 2> gpurun_out/bench_b8192.err; echo "rc=$?"; cat gpurun_out/bench_b8192.json
echo "== rocprof"; cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$OLDPWD/gpurun_out/prof_r1" -o r1 -- python "$OLDPWD/bench.py" --steps 10 --warmup 3 --no-cpu-baseline > "$OLDPWD/gpurun_out/rocprof.log" 2>&1; echo "rocprof rc=$?"; cd "$OLDPWD"
find gpurun_out/prof_r1 -name "*stats*" | head; for f in $(find gpurun_out/prof_r1 -name "*kernel_stats*.csv" | head -1); do head -25 "$f"; done
