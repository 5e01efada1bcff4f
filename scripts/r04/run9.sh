#!/bin/bash
# round 4: the whole default line (config models, int32 wire e2e, proxy child first) + the graph pipeline test
set -u
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
TAG=${1:-r04p}
O=gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_graph_pipeline_gpu.py tests/test_interaction_top.py tests/test_sharded_gpu.py -m gpu -x -q 2>&1 | tail -5 ) > $O/gpu_tests.log; echo "tests: $(grep -h "passed\|failed" $O/gpu_tests.log | tail -1)"
timeout 1500 python bench.py --steps 20 --warmup 5 2> $O/bench_default.err | tail -1 > $O/bench_default.json; echo "bench rc=$?"
python - <<PY
import json
d=json.load(open("$O/bench_default.json"))
print("value %.1f M  %.4f ms" % (d["value"]/1e6, d["ms_per_step"]), "frac", round(d["roofline"]["frac"],4))
s=d["secondary"]
for k in ("config2_batch8192","deepfm_criteo_b8192","din_taobao_b8192","mmoe_zch_b8192","sharded_w1_proxy_b8192"):
    v=s.get(k,{})
    print(k, {a: (round(b,4) if isinstance(b,float) else b) for a,b in v.items() if a in ("ms_per_step","graph_ms_per_step","host_queue_ms_per_step","error","graph_error","zch","ids_per_step")})
print("proxy scaling", s.get("sharded_w1_proxy_b8192",{}).get("projection",{}).get("scaling_vs_n1"))
print("e2e", d["e2e"]["ms_per_step"], d["e2e"]["graph_ms_per_step_runs"], d["e2e"]["h2d_bytes_per_step"], "eager", d["e2e"]["eager_ms_per_step"])
PY
tail -5 $O/bench_default.err
