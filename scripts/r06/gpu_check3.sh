#!/bin/bash
# fused dense Adam (gradients as partial sums) + the one-launch index plan: tests, default line, its A/B switches, kernel trace
set -u
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
TAG=${1:-r06k}
O=$PWD/gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q > $O/gpu_tests.log 2>&1; tail -2 $O/gpu_tests.log
timeout 900 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
timeout 600 python bench.py --no-cpu-baseline --no-secondary --no-e2e --no-fuse-finish > $O/bench_no_fuse_finish.json 2>> $O/bench.err
timeout 600 python bench.py --no-cpu-baseline --no-secondary --no-e2e --global-batch 8192 > $O/bench_b8192.json 2>> $O/bench.err
timeout 600 python bench.py --no-cpu-baseline --no-secondary --no-e2e --global-batch 8192 --no-fuse-finish > $O/bench_b8192_no_fuse_finish.json 2>> $O/bench.err
python - <<PY
import json
for f in ("bench","bench_no_fuse_finish","bench_b8192","bench_b8192_no_fuse_finish"):
    d=json.load(open("$O/%s.json" % f)); r=d.get("roofline") or {}
    print(f, round(d["ms_per_step"],4), round(r.get("frac",0),4), [(k["stage"][:14], round(k["launch_ms"]*1e3,1)) for k in r.get("kernels",[])])
d=json.load(open("$O/bench.json")); s=d.get("secondary") or {}
for k,v in s.items():
    if isinstance(v, dict): print(k, {kk: (round(vv,4) if isinstance(vv,float) else vv) for kk,vv in v.items() if kk in ("ms_per_step","graph_ms_per_step")}, (v.get("embedding") or {}).get("frac_of_8TBps"))
print("e2e", (d.get("e2e") or {}).get("ms_per_step"))
PY
R=$PWD; cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace -o t --output-format csv -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-e2e --no-graph --no-secondary > $O/trace.log 2>&1; echo "trace rc=$?"
S=$(find $O/trace -name '*kernel_stats.csv' | head -1); cp "$S" $O/kernel_stats.csv; rm -rf $O/trace
grep "tzr_" $O/kernel_stats.csv | cut -d, -f1-4 | cut -c1-60,100-200 | head -20
