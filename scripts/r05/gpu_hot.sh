#!/bin/bash
# the cooperative hot row of tzr_pooled_bwd_direct: GPU parity, the uniform-id case it must not slow down (batch 8192 step and its
# embedding stages), and the step it is for (MMoE + zero-collision hash: 95 % of the user ids read the shared row)
set -u
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
TAG=${1:-r05ab}; R=$PWD; O=$R/gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_pooled_parity.py tests/test_zch_parity.py tests/test_dense_glue.py tests/test_graph_pipeline_gpu.py -m gpu -x -q > $O/gpu_tests.log 2>&1; echo "tests rc=$?"; tail -4 $O/gpu_tests.log | cut -c1-300
for rep in 1 2; do
timeout 300 python bench.py --global-batch 8192 --steps 200 --no-cpu-baseline --no-e2e --no-secondary > $O/bench_b8192.$rep.json 2>> $O/bench.err
python - $O/bench_b8192.$rep.json <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print("b8192 ms_per_step", round(d["ms_per_step"],4), {k: round(v,4) if isinstance(v,float) else v for k,v in d["embedding"].items()})
PY
done
timeout 900 python scripts/r05/models_step.py 30 mmoe_zch_b8192 > $O/models_step.txt 2>&1; grep '"model"' $O/models_step.txt | cut -c1-900 || tail -20 $O/models_step.txt
