#!/bin/bash
# round 4: (a) B = 65536 embedding stages with the branch-free plan kernels, (b) kernel trace of the 8192-per-rank sharded proxy
set -u
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
TAG=${1:-r04m}
O=gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
R=$PWD
for i in 1 2; do
timeout 600 python bench.py --steps 20 --warmup 5 --no-e2e --no-cpu-baseline --no-secondary 2> $O/b65536.err | tail -1 > $O/b65536_$i.json
python - <<PY
import json
d=json.load(open("$O/b65536_$i.json")); r=d["roofline"]
print("B=65536 run $i: step %.4f ms frac %.4f" % (d["ms_per_step"], r["frac"]), [(k["stage"][:14], round(k["launch_ms"]*1e3,1)) for k in r["kernels"]])
PY
done
( timeout 900 python -m pytest tests/test_pooled_parity.py tests/test_plan_invariants.py tests/test_fullsize_properties.py -m gpu -x -q 2>&1 | tail -3 ) > $O/gpu_tests_plan.log; echo "plan tests: $(grep -h passed $O/gpu_tests_plan.log | tail -1)"
cd /tmp
timeout 400 rocprofv3 --kernel-trace -d "$R/$O/trace" -o t -- python "$R/bench.py" --force-sharded --replicate-small --global-batch 8192 --steps 20 --warmup 12 --no-cpu-baseline > "$R/$O/trace.log" 2>&1; echo "trace rc=$?"
cd "$R"
DB=$(find $O/trace -name '*.db' | head -1)
python scripts/rocpd_stats.py "$DB" $O/kernel_stats_proxy8192.csv
python scripts/rocpd_timeline.py "$DB" 260 $O/timeline_proxy8192.txt
rm -rf $O/trace
